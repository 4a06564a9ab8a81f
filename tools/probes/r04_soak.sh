#!/bin/bash
# final soak of round 4: every resident form hammered against the multi-launch schedule (bit for bit)
O=gpurun_out/soak; mkdir -p $O
timeout 1200 python tools/probes/resident_stress.py 30000 > $O/res_plain.txt 2>&1; tail -4 $O/res_plain.txt
timeout 1200 python tools/probes/resident_stress.py 30000 sparse > $O/res_sparse.txt 2>&1; tail -4 $O/res_sparse.txt
timeout 1200 python tools/probes/kres_stress.py 6000 > $O/kres.txt 2>&1; tail -4 $O/kres.txt
timeout 1200 python tools/probes/resident_train_fuzz.py 120 > $O/train_fuzz.txt 2>&1; tail -2 $O/train_fuzz.txt
