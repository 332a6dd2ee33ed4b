"""Reference module path `text_model` (anthonyhu/tumblr-emotions): with <repo>/compat and <repo> on PYTHONPATH the
reference's callers (parallel_computing/job_train.py:4-7, job_evaluate.py:3-5, job_*.py) import unchanged; every
submodule here IS the module of the same name under tumblr_emotions_amd.text_model (one module object, not a copy)."""
