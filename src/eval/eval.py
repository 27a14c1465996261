#!/usr/bin/env python3
"""Scorer for the results JSONL written by src/eval/infer.py: drop-in for the reference's `src/eval/eval.py`
(/root/reference/src/eval/eval.py:44-152).

Per record: a stage answer is correct when it equals the ground truth (case-insensitive, stripped) or is a WordNet
synonym of it (best path similarity >= 0.8 over all synset pairs of the lemmatised words).  `answer2 is None`
falls back to `answer1`.  Accuracy is reported overall and per `type`, stage 1 -> stage 2.  When `nltk` / the
WordNet corpus is not installed the synonym test runs on a bundled subset of WordNet 3.0 synsets
(src/eval/synsets_lite.json; pinned by tests/golden/synonym_cases.json) and the report says so.
"""
import argparse
import json
from collections import defaultdict

import os

try:  # WordNet synonyms through nltk when it (and its corpus) is installed
    import nltk  # noqa: F401
    from nltk.corpus import wordnet as _wn
    from nltk.stem import WordNetLemmatizer as _Lem
    _wn.synsets("tree")
    HAVE_WORDNET = True
except Exception:  # pragma: no cover - nltk is absent in the build image
    HAVE_WORDNET = False

# Without nltk: a bundled, hand-transcribed subset of WordNet 3.0 (src/eval/synsets_lite.json).  The reference's rule
# (/root/reference/src/eval/eval.py:22-42) -- best path_similarity over all synset pairs >= 0.8 -- is equivalent to
# "the two lemmatised words share a synset": path_similarity = 1 / (1 + hypernym-path distance) is 1 for an identical
# synset and at most 1/2 otherwise, so only membership tables are needed, no hypernym graph.
_LITE = None


def _lite():
    global _LITE
    if _LITE is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synsets_lite.json"), encoding="utf-8") as f:
            d = json.load(f)
        index = defaultdict(set)
        for name, lemmas in d["synsets"].items():
            for lemma in lemmas:
                index[lemma].add(name)
        _LITE = (index, d.get("noun_plurals", {}))
    return _LITE


def _lemmatize_lite(word: str) -> str:
    """WordNetLemmatizer().lemmatize(word) (noun) on the bundled vocabulary: the exception / plural table first, then
    morphy's noun detachment rules, accepted only when the result is a known lemma; otherwise the word unchanged."""
    index, plurals = _lite()
    if word in index:
        return word
    if word in plurals:
        return plurals[word]
    for suffix, repl in (("ses", "s"), ("ves", "f"), ("xes", "x"), ("zes", "z"), ("ches", "ch"), ("shes", "sh"),
                         ("men", "man"), ("ies", "y"), ("s", "")):
        if word.endswith(suffix) and word[: len(word) - len(suffix)] + repl in index:
            return word[: len(word) - len(suffix)] + repl
    return word


def are_synonyms(a: str, b: str) -> bool:
    if a is None or b is None:
        return False
    if not HAVE_WORDNET:
        index, _ = _lite()
        # (no space -> underscore mapping: nltk's wordnet.synsets() has none, so "parking lot" finds nothing there either)
        s1 = index.get(_lemmatize_lite(a.lower()), set())
        s2 = index.get(_lemmatize_lite(b.lower()), set())
        return bool(s1 & s2)
    try:
        lem = _Lem()
        s1, s2 = _wn.synsets(lem.lemmatize(a.lower())), _wn.synsets(lem.lemmatize(b.lower()))
    except Exception:
        return False
    best = 0.0
    for x in s1:
        for y in s2:
            sim = x.path_similarity(y)
            if sim is not None and sim > best:
                best = sim
    return best >= 0.8


def _norm(ans):
    return ans.lower().strip() if ans is not None else None


def score_records(records):
    """Returns dict(total, correct1, correct2, by_type={type: (n, c1, c2)}, fixed, broken)."""
    by_type = defaultdict(lambda: [0, 0, 0])
    c1 = c2 = 0
    fixed, broken = [], []
    for item in records:
        gt = (item.get("ground_truth") or "").lower()
        a1 = _norm(item.get("answer1"))
        a2 = _norm(item.get("answer2"))
        if a2 is None:
            a2 = a1
        ok1 = gt == a1 or are_synonyms(gt, a1)
        ok2 = gt == a2 or are_synonyms(gt, a2)
        c1 += ok1
        c2 += ok2
        t = by_type[item["type"]]
        t[0] += 1
        t[1] += ok1
        t[2] += ok2
        if ok1 and not ok2:
            broken.append(item)
        if ok2 and not ok1:
            fixed.append(item)
    return dict(total=len(records), correct1=c1, correct2=c2, by_type={k: tuple(v) for k, v in by_type.items()},
                fixed=fixed, broken=broken)


def evaluation_metrics(data_path):
    with open(data_path, encoding="utf-8") as f:
        records = [json.loads(line) for line in f if line.strip()]
    r = score_records(records)
    n = r["total"]
    print("\n" + "=" * 50 + "\nEvaluating dataset: LRS-GRO\n" + "=" * 50)
    print("Processing evaluations...")
    if not HAVE_WORDNET:
        import sys
        print("(nltk/WordNet not installed: synonym credit from the bundled WordNet subset src/eval/synsets_lite.json only)",
              file=sys.stderr)  # (stderr: stdout is the reference's report, line for line -- tests/golden/eval_report.json)
    print("\n--- Evaluation Results ---")
    print(f"Total Correct (stage 1): {r['correct1']}")
    print(f"Total Correct (stage 2): {r['correct2']}")
    print(f"Total Incorrect (stage 1): {n - r['correct1']}")
    print(f"Total Incorrect (stage 2): {n - r['correct2']}")
    print(f"Total Samples: {n}")
    # (the reference prints an empty category section, /root/reference/src/eval/eval.py:105-107: kept, line for line)
    print("-" * 25 + "\nCategory-wise Accuracies:\n" + "-" * 25 + "\nType-wise Accuracies:")
    for t in sorted(r["by_type"]):
        cnt, a, b = r["by_type"][t]
        print(f"{t:<15}: {100.0 * a / cnt:.2f}% -> {100.0 * b / cnt:.2f}%")
    print("-" * 25)
    if n:
        print(f"Overall Accuracy (OA, stage 1): {100.0 * r['correct1'] / n:.2f}%")
        print(f"Overall Accuracy (OA, stage 2): {100.0 * r['correct2'] / n:.2f}%")
        # the reference's counter `total_correct_call` is never incremented (:49, :124): the line always reads 0.0000 %
        print(f"Tool use accuracy: {100.0 * r.get('correct_call', 0) / n:.4f}%")
        print(f"Overall: {100.0 * r['correct1'] / n:.2f}% -> {100.0 * r['correct2'] / n:.2f}%")
    else:
        print("Overall Accuracy (OA): N/A (No samples found)")
    return r


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--results_file", type=str, default="")
    evaluation_metrics(parser.parse_args().results_file)
