"""CPU: the scorer's synonym rule (/root/reference/src/eval/eval.py:22-42) pinned by hand-derived WordNet cases.

`are_synonyms(a, b)` = best path_similarity over all synset pairs of the lemmatised words >= 0.8.  Since
path_similarity = 1 / (1 + path distance) in {1, 1/2, 1/3, ...}, the rule holds exactly when the words share a synset;
the build image has no nltk, so src/eval/eval.py falls back to a bundled synset subset and these cases pin both the
rule and the table.  Where nltk + its corpus exist, the same cases run against the real WordNet."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_scorer():
    spec = importlib.util.spec_from_file_location("ze_eval", os.path.join(ROOT, "src", "eval", "eval.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def cases():
    with open(os.path.join(ROOT, "tests", "golden", "synonym_cases.json"), encoding="utf-8") as f:
        return json.load(f)["cases"]


def test_synonym_cases_on_the_scorer_as_installed():
    m = load_scorer()
    assert len(cases()) >= 15
    for a, b, want, why in cases():
        assert m.are_synonyms(a, b) is want, (a, b, why)
        assert m.are_synonyms(b, a) is want, (b, a, why)   # the rule is symmetric
    assert m.are_synonyms("car", None) is False and m.are_synonyms(None, "car") is False
    assert m.are_synonyms("zzzunknown", "car") is False


def test_cases_against_nltk():
    """Only where nltk and the WordNet corpus are installed: the hand-derived expectations against the real database."""
    pytest.importorskip("nltk")
    m = load_scorer()
    if not m.HAVE_WORDNET:
        pytest.skip("WordNet corpus not installed")
    for a, b, want, why in cases():
        assert m.are_synonyms(a, b) is want, (a, b, why)


def test_scoring_uses_synonyms_and_stage_fallback():
    m = load_scorer()
    recs = [dict(ground_truth="Car", answer1="automobile", answer2=None, type="object"),        # synonym, answer2 falls back
            dict(ground_truth="ship", answer1="vessel", answer2="ship ", type="object"),        # hypernym is wrong; stage 2 exact
            dict(ground_truth="3", answer1="3", answer2="4", type="count"),                      # broken by stage 2
            dict(ground_truth="woods", answer1=None, answer2="Forest", type="scene")]            # None answer1; synonym in stage 2
    r = m.score_records(recs)
    assert (r["total"], r["correct1"], r["correct2"]) == (4, 2, 3)
    assert r["by_type"] == {"object": (2, 1, 2), "count": (1, 1, 0), "scene": (1, 0, 1)}
    assert len(r["fixed"]) == 2 and len(r["broken"]) == 1
