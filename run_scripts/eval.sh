#!/usr/bin/env bash
# Drop-in for the reference's run_scripts/eval.sh: scores a results JSONL (exact match + WordNet synonyms when nltk is
# installed; exact match only otherwise).
file_path="${file_path:-${1:-}}"
echo "Evaluating inference file: $file_path!"
python src/eval/eval.py --results_file "$file_path"
