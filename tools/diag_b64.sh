cd $GRAFT_REPO_ROOT
echo "== default"; python3 tools/diag_block64.py 2>&1 | grep "^run"
echo "== -fno-slp-vectorize"; PPN_LIB=tools/bin/libppn_b64noslp.so python3 tools/diag_block64.py 2>&1 | grep "^run"
