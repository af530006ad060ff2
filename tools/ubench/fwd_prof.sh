#!/bin/bash
cd $GRAFT_REPO_ROOT
PROF=1 SLOTS=8 FWD=1 REPS=20 timeout 60 tools/ubench/bwd_ab.bin tools/ubench/libyunet_fprof.so 2>&1 | grep -E "80x80|40x40|clocks" | head -4
