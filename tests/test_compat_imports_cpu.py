"""The reference's callers import unchanged (SURVEY 8b upper face): with <repo>/compat on the path, exactly the import
lines of /root/reference/parallel_computing/job_train.py:4-7, job_evaluate.py:3-5, job_correlation.py:2,
job_outliers.py:1, job_week.py:2, job_top_words.py:6-8 resolve -- to the SAME objects as the tumblr_emotions_amd modules."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLER_IMPORTS = """
from image_model.im_model import download_pretrained_model
from image_model.im_model import train_image_model
from text_model.text_embedding import train_text_model
from image_text_model.im_text_rnn_model import train_deep_sentiment
from image_model.im_model import evaluate_image_model
from text_model.text_embedding import evaluate_text_model
from image_text_model.im_text_rnn_model import evaluate_deep_sentiment
from image_text_model.im_text_rnn_model import correlation_matrix
from image_text_model.im_text_rnn_model import outliers_detection
from image_text_model.im_text_rnn_model import day_of_week_trend
from text_model.text_preprocessing import _load_embedding_weights_glove, preprocess_one_df
from image_text_model.im_text_rnn_model import word_most_relevant
from datasets.dataset_utils import read_label_file
from image_model.inception_v1 import inception_v1, inception_v1_base, inception_v1_arg_scope, default_image_size
"""

CHECK = CALLER_IMPORTS + """
import image_model.im_model, text_model.text_embedding, image_text_model.im_text_rnn_model
import tumblr_emotions_amd.image_model.im_model as a
import tumblr_emotions_amd.text_model.text_embedding as b
import tumblr_emotions_amd.image_text_model.im_text_rnn_model as c
assert image_model.im_model is a and text_model.text_embedding is b and image_text_model.im_text_rnn_model is c
assert train_deep_sentiment is c.train_deep_sentiment and train_image_model is a.train_image_model
assert default_image_size == 224
print("compat ok")
"""


def test_reference_callers_import_unchanged():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]), PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", CHECK], env=env, cwd="/tmp", capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "compat ok" in r.stdout, r.stderr[-2000:]
