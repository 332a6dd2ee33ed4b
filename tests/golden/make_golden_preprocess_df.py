#!/usr/bin/env python
"""preprocess_df.json: a small synthetic ./data/<emotion>.csv set + GloVe file, and what the REFERENCE's own
text_model/text_preprocessing.py (preprocess_df :107-143, preprocess_one_df :145-172) returns for it -- captured by
importing the reference module in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_preprocess_df.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

COLUMNS = ['id', 'post_url', 'type', 'timestamp', 'date', 'tags', 'liked', 'note_count', 'photo', 'text', 'search_query']
GLOVE = ["the 0.5 -1.25", "happy 1 2", "dog 3 4", "is 0.25 0.5", "a -1 -2", "very 5 6", "good 7 8", "boy 9 10",
         "sun 0.125 0", "sad -3 3", "rain 2 -2", "day 1.5 2.5", "today 4 -4", "walk 6 -6"]
LONG = u"The happy dog is a very good boy #happy"
POSTS = {
    "happy": [
        (1, "[happy, sun, dog]", LONG),
        (2, "[sun, dog]", LONG),                                          # hashtag missing from tags: dropped
        (3, "[happy]", u"the dog"),                                       # too few vocabulary words: dropped
        (4, "[ happy ,walk]", u"Today, the SUN is very good!! a walk day #sad #happy unknownword"),
        (5, "[happy]", None),                                             # NaN text: dropped
    ],
    "sad": [
        (6, "[sad, rain]", u"the rain today is a very sad day #sad #surprised"),
        (7, "[sad]", u"sad sad sad sad sad sad sad"),                      # one distinct word: dropped
        (8, "[rain, sad]", u"a boy, a dog; the rain: is very good? the sun (today) walk walk walk walk walk walk"),
    ],
}


def write_inputs(root):
    os.makedirs(os.path.join(root, "data"))
    os.makedirs(os.path.join(root, "text", "emb"))
    with open(os.path.join(root, "text", "emb", "g.txt"), "w") as f:
        f.write("\n".join(GLOVE) + "\n")
    import pandas as pd
    for emotion, posts in POSTS.items():
        rows = [[i, "http://x/%d" % i, "photo", 1500000000 + i, "2017-07-0%d" % (i % 9 + 1), tags, False, i * 3,
                 "http://x/%d.jpg" % i, text, emotion] for (i, tags, text) in posts]
        pd.DataFrame(rows, columns=COLUMNS).to_csv(os.path.join(root, "data", emotion + ".csv"), index=False, encoding="utf-8")


def frame_to_json(df):
    return dict(id=[int(v) for v in df["id"]], text_list=[[int(i) for i in l] for l in df["text_list"]],
                text_len=[int(v) for v in df["text_len"]], search_query=[v if isinstance(v, str) else int(v) for v in df["search_query"]],
                tags=[sorted(t) for t in df["tags"]])


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    from text_model import text_preprocessing as ref
    emotions = ["happy", "sad"]
    out = dict(columns=COLUMNS, glove=GLOVE, posts=POSTS, emotions=emotions, cases=[])
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        write_inputs(d)
        os.chdir(d)                      # the reference reads ./data/<emotion>.csv
        try:
            for post_size in (6, 12):
                df, w2i, emb = ref.preprocess_df(os.path.join(d, "text"), "emb", "g.txt", "glove", emotions, post_size)
                case = dict(post_size=post_size, all=frame_to_json(df), word_to_id=w2i,
                            embedding=np.asarray(emb, np.float64).tolist(), one={})
                v, e = ref._load_embedding_weights_glove(os.path.join(d, "text"), "emb", "g.txt")
                for emotion in emotions:
                    case["one"][emotion] = frame_to_json(ref.preprocess_one_df(v, e, emotion, post_size))
                out["cases"].append(case)
        finally:
            os.chdir(cwd)
    with open(os.path.join(HERE, "preprocess_df.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote preprocess_df.json:", [len(c["all"]["id"]) for c in out["cases"]], "rows")


if __name__ == "__main__":
    main()
