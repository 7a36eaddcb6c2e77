#!/bin/bash
export TMPDIR=/tmp
echo coset; timeout 300 python tools/lde_time.py 24 1 6 19 36 2>/dev/null
echo old; ZK_LDE_NO_COSET=1 timeout 300 python tools/lde_time.py 24 1 6 19 36 2>/dev/null
echo coset 2^20; timeout 300 python tools/lde_time.py 20 19 36 2>/dev/null
echo old 2^20; ZK_LDE_NO_COSET=1 timeout 300 python tools/lde_time.py 20 19 36 2>/dev/null
