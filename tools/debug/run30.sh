cd $GRAFT_REPO_ROOT
for l in 0 54000 65000 0 54000 65000; do VCLA_ATTN_LDS=$l python tools/pmc_attn_decode.py 2>&1 | tail -1 | awk -v w=$l '{print "lds" w, $0}'; done
