import numpy as np
from scipy import ndimage
def label(x, background=0, connectivity=None):
    return ndimage.label(np.asarray(x) != background, structure=np.ones((3, 3)))[0]
