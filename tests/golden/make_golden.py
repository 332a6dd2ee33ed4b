#!/usr/bin/env python
"""Generate the committed golden vectors.  Run in the BUILD container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

(1) text_preprocessing.json -- inputs and outputs of the reference's OWN tokeniser
    (/root/reference/text_model/text_preprocessing.py, the only reference module importable here),
    captured by importing it.  These are real reference outputs.
(2) joint_step_oracle.npz -- one joint training step (B=2) computed by the fp64 PyTorch-CPU oracle
    (TensorFlow is not installable, so this is an oracle regression vector, NOT a reference output):
    seeds + expected logits / loss / gradient checksums, so the GPU box can check the HIP path against
    numbers that were fixed at build time.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def text_preprocessing_vectors():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from text_model import text_preprocessing as ref
    vocab = ["the", "happy", "dog", "is", "a", "very", "good", "boy", "sun", "co-op", "@you", "don't", "dont", "café"]
    w2i = dict(zip(vocab, range(len(vocab))))
    emotions = ["happy", "sad", "happyness"]
    paragraphs = [
        u"The #happy Dog!! is, the unknownword",
        u"",
        u"   ",
        u"#sad #sad #SAD the dog",
        u"Don't stop: the co-op @you (very) good_boy... is a GOOD boy? #happyness",
        u"café CAFÉ café; the\tsun\nis   a   very very very good good boy boy dog dog the the the end",
        u"a b c d e f g h i j k l m n o p q r s t u v w x y z the the",
        u"the happy dog is a very good boy sun",
        u"#happythe dog",
        u"the.dog,is;a:very!good?boy",
    ]
    cases = []
    for post_size in (8, 50):
        for p in paragraphs:
            ids, n = ref._paragraph_to_ids(p, w2i, post_size, emotions)
            cases.append(dict(paragraph=p, post_size=post_size, ids=[int(i) for i in ids], length=int(n)))
    valid = [dict(paragraph=p, valid=bool(ref._is_valid_text(p, set(vocab)))) for p in paragraphs]
    valid.append(dict(paragraph=None, valid=bool(ref._is_valid_text(float("nan"), set(vocab)))))
    sets = [dict(text=s, items=sorted(ref._str_list_to_set(s))) for s in
            ["[happy, sun, outdoors]", "[a]", "[]", "[ x ,y,  z z ]"]]
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "emb"))
        rows = ["the 0.5 -1.25 3e-2", "dog 1 2 3", "café -0.001 0 7.5"]
        with open(os.path.join(d, "emb", "g.txt"), "w") as f:
            f.write("\n".join(rows) + "\n")
        v, e = ref._load_embedding_weights_glove(d, "emb", "g.txt")
    glove = dict(rows=rows, vocabulary=list(v), embedding=np.asarray(e, np.float64).tolist(), dtype=str(np.asarray(e).dtype))
    return dict(vocabulary=vocab, emotions=emotions, paragraph_to_ids=cases, is_valid_text=valid,
                str_list_to_set=sets, glove=glove, punctuation=ref._PUNCTUATION,
                source="/root/reference/text_model/text_preprocessing.py (imported, outputs captured)")


def joint_step_vectors():
    import torch
    from oracle import tf_semantics as S
    from oracle import torch_ref as R
    cfg = dict(V=50, D=16, H=32, T=8, B=2, param_seed=0, batch_seed=0, lr=1e-3)
    rng = np.random.RandomState(cfg["param_seed"])
    params = R.make_params("joint", rng, num_classes=15, im_features_size=256, embed_dim=cfg["D"], rnn_size=cfg["H"],
                           fc_size=512, dtype=np.float64)
    emb = S.synthetic_embedding(cfg["V"], cfg["D"]).astype(np.float64)
    batch = S.synthetic_batch(cfg["B"], cfg["T"], cfg["V"], seed=cfg["batch_seed"])
    mask = (rng.uniform(size=(cfg["B"], 1024)) < 0.8).astype(np.float64)
    ref = R.DeepSentimentRef(params, emb, "joint", torch.float64)
    out = ref.train_step(batch, cfg["lr"], torch.tensor(mask))
    grads = {k: v.numpy() for k, v in out["grads"].items()}
    keep = ["InceptionV1/Logits/Conv2d_0c_1x1/weights", "InceptionV1/Logits/Conv2d_0c_1x1/biases",
            "Text/rnn/basic_lstm_cell/kernel", "Text/rnn/basic_lstm_cell/bias", "W_fc", "b_fc", "W_softmax", "b_softmax"]
    np.savez_compressed(os.path.join(HERE, "joint_step_oracle.npz"), cfg=json.dumps(cfg),
                        logits=out["logits"].numpy(), loss=out["loss"], ce=out["ce"], dropout_mask=mask.astype(np.float32),
                        **{"grad/" + k: grads[k].astype(np.float32) for k in keep})


if __name__ == "__main__":
    with open(os.path.join(HERE, "text_preprocessing.json"), "w") as f:
        json.dump(text_preprocessing_vectors(), f, indent=1, ensure_ascii=True)
    joint_step_vectors()
    print("golden vectors written to", HERE)
