"""Imported first by the scripts that pin tiles / kernel families (ds_debug_*) or sweep the DS_* environment knobs of the
selection rules: points the package at the -DDS_TUNING build (libds_kernels_tuning.so).  The shipped libds_kernels.so has
neither (include/ds_kernels.h)."""
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DS_LIB", os.path.join(_ROOT, "tumblr_emotions_amd", "libds_kernels_tuning.so"))
