"""ccd_amd - MI355X-native implementation of the CCD self-supervised pretraining step.

The HIP library (ccd_amd/csrc -> libccd_hip.so) is loaded lazily by ccd_amd._lib on first use of any op and
raises loudly if it is missing: there is no CPU fallback in this package.
"""
__version__ = "0.1.0"
