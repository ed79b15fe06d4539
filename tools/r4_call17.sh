#!/bin/bash
# class-sequence tile kernel after the scalar-instruction cuts: its tests, then the stage alone at 1 GiB
O=gpurun_out/r4q; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "class or seq or facade" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python tools/seq_prof.py 1024 > $O/seq.log 2>&1
