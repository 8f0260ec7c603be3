"""CPU tests: the oracle's SLOT LAYOUT against the survey-time independent restatement
(SURVEY.md App. B).  The reference's own tests do not pin slot positions; these digests
are the only second opinion available without a Julia toolchain."""
import json
import os

import ka

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "survey_known_answers.json")) as f:
    KA = json.load(f)


def subset_equal(got, exp, path=""):
    for k, v in exp.items():
        assert k in got, path + k
        if isinstance(v, dict):
            subset_equal(got[k], v, path + k + ".")
        else:
            assert got[k] == v, (path + k, got[k], v)


def test_ka4_ascending_inserts(dsa, oracle):
    subset_equal(ka.ka4(dsa, oracle), KA["ka4"])


def test_ka5_descending_inserts(dsa, oracle):
    subset_equal(ka.ka5(dsa, oracle), KA["ka5"])


def test_ka6_random_fill_and_empty(dsa, oracle):
    subset_equal(ka.ka6(dsa, oracle), KA["ka6"])


def test_ka6_batched_equals_sequential(dsa, oracle):
    got = ka.ka6(dsa, oracle, batch=True)
    exp = json.loads(json.dumps(KA["ka6"]))
    exp["c"].pop("shrinks")
    subset_equal(got, exp)


def test_ka7_to_ka9_matrix(dsa, oracle):
    got = ka.ka7_9(dsa, oracle)
    subset_equal(got, {k: KA[k] for k in ("ka7", "ka8", "ka9")})
    assert got["ka7"]["y4_sparse"] == KA["ka7"]["y4"]
