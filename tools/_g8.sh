cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
D=gpurun_out/r02c
# plain kernels (no sc1 state accesses), envs hop between XCDs from step to step, with and without the packets' release fence
SSD_TEST_ALT_PLAIN=1 SSD_AQL_ALTERNATE=1 SSD_AQL_REL=1 SSD_AQL_ACQ=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/exp_plain_alt_rel1.log 2>&1; echo "plain alt rel=agent acq=agent rc=$?"; grep -i "assert\|error" $D/exp_plain_alt_rel1.log | tail -2
SSD_TEST_ALT_PLAIN=1 SSD_AQL_ALTERNATE=1 SSD_AQL_REL=0 SSD_AQL_ACQ=1 timeout -k 10 150 python3 tools/aql_smoke.py > $D/exp_plain_alt_rel0.log 2>&1; echo "plain alt rel=none acq=agent rc=$?"; grep -i "assert\|error" $D/exp_plain_alt_rel0.log | tail -2
SSD_TEST_ALT_PLAIN=1 SSD_AQL_ALTERNATE=1 SSD_AQL_REL=1 SSD_AQL_ACQ=0 timeout -k 10 150 python3 tools/aql_smoke.py > $D/exp_plain_alt_acq0.log 2>&1; echo "plain alt rel=agent acq=none rc=$?"; grep -i "assert\|error" $D/exp_plain_alt_acq0.log | tail -2
SSD_TEST_ALT_PLAIN=1 SSD_AQL_ALTERNATE=1 SSD_AQL_REL=0 SSD_AQL_ACQ=0 timeout -k 10 150 python3 tools/aql_smoke.py > $D/exp_plain_alt_none.log 2>&1; echo "plain alt rel=none acq=none rc=$?"; grep -i "assert\|error" $D/exp_plain_alt_none.log | tail -2
