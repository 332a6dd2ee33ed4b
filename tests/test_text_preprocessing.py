"""tokeniser / GloVe loader vs vectors captured from the reference's own module
(tests/golden/text_preprocessing.json, produced by tests/golden/make_golden.py importing
/root/reference/text_model/text_preprocessing.py)."""
import json
import os

import numpy as np

from tumblr_emotions_amd.text_model import text_preprocessing as tp

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_preprocessing.json")))


def test_punctuation_set_is_the_references():
    assert tp._PUNCTUATION == G["punctuation"]
    assert "-" not in tp._PUNCTUATION and "@" not in tp._PUNCTUATION


def test_paragraph_to_ids_matches_reference_outputs():
    w2i = dict(zip(G["vocabulary"], range(len(G["vocabulary"]))))
    assert len(G["paragraph_to_ids"]) == 20
    for case in G["paragraph_to_ids"]:
        ids, n = tp._paragraph_to_ids(case["paragraph"], w2i, case["post_size"], G["emotions"])
        assert ids == case["ids"] and n == case["length"], case["paragraph"]
        assert len(ids) == case["post_size"]
        assert all(i == len(w2i) for i in ids[n:])          # pad id = unk id = vocabulary size


def test_is_valid_text_matches_reference_outputs():
    vocab = set(G["vocabulary"])
    for case in G["is_valid_text"]:
        p = float("nan") if case["paragraph"] is None else case["paragraph"]
        assert tp._is_valid_text(p, vocab) == case["valid"], case["paragraph"]


def test_str_list_to_set_matches_reference_outputs():
    for case in G["str_list_to_set"]:
        assert sorted(tp._str_list_to_set(case["text"])) == case["items"]


def test_glove_loader_matches_reference_outputs(tmp_path):
    os.makedirs(tmp_path / "emb")
    (tmp_path / "emb" / "g.txt").write_text("\n".join(G["glove"]["rows"]) + "\n")
    vocab, emb = tp._load_embedding_weights_glove(str(tmp_path), "emb", "g.txt")
    assert vocab == G["glove"]["vocabulary"]
    assert str(emb.dtype) == G["glove"]["dtype"]
    np.testing.assert_array_equal(emb.astype(np.float64), np.array(G["glove"]["embedding"]))
    w2i, table = tp.build_vocabulary(vocab, emb)
    assert w2i["<ukn>"] == len(vocab) and table.shape == (len(vocab) + 1, 3) and not table[-1].any()


def test_preprocess_df_and_one_df_match_reference_outputs(tmp_path):
    """preprocess_df (:107-143) / preprocess_one_df (:145-172) on the synthetic ./data set of
    tests/golden/make_golden_preprocess_df.py against what the reference module returned for it."""
    import importlib.util
    P = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "preprocess_df.json")))
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden",
                                                                     "make_golden_preprocess_df.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    mk.write_inputs(str(tmp_path))
    data_dir, text_dir = str(tmp_path / "data"), str(tmp_path / "text")
    assert len(P["cases"]) == 2
    for case in P["cases"]:
        df, w2i, emb = tp.preprocess_df(text_dir, "emb", "g.txt", "glove", P["emotions"], case["post_size"], data_dir=data_dir)
        assert mk.frame_to_json(df) == case["all"]
        assert w2i == case["word_to_id"] and w2i["<ukn>"] == len(P["glove"])
        np.testing.assert_array_equal(np.asarray(emb, np.float64), np.asarray(case["embedding"]))
        assert not np.asarray(emb)[-1].any()
        v, e = tp._load_embedding_weights_glove(text_dir, "emb", "g.txt")
        for emotion in P["emotions"]:
            one = tp.preprocess_one_df(v, e, emotion, case["post_size"], data_dir=data_dir)
            assert mk.frame_to_json(one) == case["one"][emotion]
    assert len(case["all"]["id"]) == 4          # four of the eight posts survive the hashtag / validity filters
