cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
