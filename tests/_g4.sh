cd $GRAFT_REPO_ROOT
python tests/debug_bn2.py 2>&1 | tail -50
