cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
