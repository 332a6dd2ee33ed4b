"""CPU oracle for the Deep Sentiment training path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

PARITY STATUS.  The reference's arithmetic lives in TensorFlow 1.x (not vendored in
/root/reference, not installable here), so this is a *restatement*, not the reference.
It is pinned against every numeric known answer the reference tree holds for this path:
  * SAME-padding conv known answers  slim/nets/resnet_v1_test.py:72-152
  * BatchNorm moving statistics      slim/deployment/model_deploy_test.py:467-524
  * Inception-v1 endpoint shapes     slim/nets/inception_v1_test.py:85-100
  * Inception-v1 variable count      slim/nets/inception_v1_test.py:109-117 (5 607 184)
  * tokeniser outputs of text_model/text_preprocessing.py (importable; golden vectors
    generated here by tests/golden/make_golden.py)
For the embedding lookup, LSTM, dynamic_rnn masking, dense head, softmax-CE + L2 and Adam the
reference holds no test, golden vector or fixture: **parity unpinned** for those (two
independent restatements, NumPy and PyTorch-CPU, are cross-checked against each other instead).
"""
