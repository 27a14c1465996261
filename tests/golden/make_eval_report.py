"""Golden output of the reference's scorer (/root/reference/src/eval/eval.py) on a small results file.

The reference imports nltk's WordNet, which is absent in the build image: it is imported here with a STUB WordNet that knows no
synsets, so the captured report pins everything of the scorer except the synonym table -- exact-match rule, lower / strip,
`answer2 is None -> answer1`, the per-type and overall lines, the empty category section, the "Tool use accuracy" line whose
counter the reference never increments -- line for line.  The records contain no pair that WordNet would call synonyms.
usage: python tests/golden/make_eval_report.py   (writes tests/golden/eval_report.json; needs /root/reference)"""
import contextlib
import importlib.util
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))

RECORDS = [
    dict(question_id=1, ground_truth="Car", answer1="car", answer2=None, type="object", category="a"),
    dict(question_id=2, ground_truth="ship", answer1="boat", answer2=" Ship ", type="object", category="a"),
    dict(question_id=3, ground_truth="3", answer1="3", answer2="4", type="count", category="b"),
    # (answer1 = None makes the reference raise AttributeError in are_synonyms -- None.lower() -- so the golden uses "")
    dict(question_id=4, ground_truth="forest", answer1="", answer2="Forest", type="scene", category="c"),
    dict(question_id=5, ground_truth="yes", answer1="no", answer2="no", type="scene", category="c"),
    dict(question_id=6, ground_truth="residential area", answer1="residential area", answer2="Residential Area", type="scene", category="c"),
    dict(question_id=7, ground_truth="north-east", answer1="northeast", answer2=None, type="direction", category="d"),
]


def main():
    wn = types.SimpleNamespace(synsets=lambda w: [])
    nltk = types.ModuleType("nltk")
    nltk.download = lambda *a, **k: True
    nltk.data = types.SimpleNamespace(path=[])
    corpus = types.ModuleType("nltk.corpus")
    corpus.wordnet = wn
    stem = types.ModuleType("nltk.stem")
    stem.WordNetLemmatizer = lambda: types.SimpleNamespace(lemmatize=lambda w, *a: w)
    nltk.corpus, nltk.stem = corpus, stem
    sys.modules.update({"nltk": nltk, "nltk.corpus": corpus, "nltk.stem": stem})
    spec = importlib.util.spec_from_file_location("ref_eval", "/root/reference/src/eval/eval.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    path = "/tmp/ze_eval_records.jsonl"
    with open(path, "w", encoding="utf-8") as f:
        for r in RECORDS:
            f.write(json.dumps(r) + "\n")
    out = io.StringIO()
    with contextlib.redirect_stdout(out), contextlib.redirect_stderr(io.StringIO()):
        m.evaluation_metrics(path)
    with open(os.path.join(HERE, "eval_report.json"), "w", encoding="utf-8") as f:
        json.dump(dict(records=RECORDS, stdout=out.getvalue(),
                       note="reference scorer run with a stub WordNet (no synsets): see make_eval_report.py"), f, indent=1)
    print(out.getvalue())


if __name__ == "__main__":
    main()
