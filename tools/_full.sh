timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
bash profiles/collect.sh 2>&1 | tail -45
