# round 6 call 19: the whole GPU suite on the final sources (the summary line was cut off in call 18)
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
