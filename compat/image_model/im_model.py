import importlib as _importlib
import sys as _sys

# the import machinery hands out sys.modules[__name__] after this file has run: the real module, private names included
_sys.modules[__name__] = _importlib.import_module("tumblr_emotions_amd.image_model.im_model")
