# round 6 call 15: the whole GPU suite + smoke() on the current sources
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -5
