#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; export TMPDIR=/tmp
(BT2G_LIB=$R/build/variants/libbt2g_diag.so timeout 300 python tools/r04_diag.py se150 1000000 2>&1 | grep -v "^\[\|amdgpu.ids" | tee $O/se.txt) | cut -c1-330
