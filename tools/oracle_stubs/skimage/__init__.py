from . import measure
