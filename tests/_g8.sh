cd $GRAFT_REPO_ROOT
python tests/ab_rgb.py pre rgb4 rgb8 cur 2>&1 | tail -6
