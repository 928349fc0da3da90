#!/bin/bash
# The parity soak of a round on the GPU box: every shape family against the CPU oracle (tests/soak_vs_oracle.py, soak_greedy_vs_oracle.py);
# census lines into gpurun_out/soak_final.txt (copied to profiles/rNN_soak.txt).
out=gpurun_out/soak_final.txt
: > $out
python tests/soak_vs_oracle.py MATE-4v8-9.yaml 4096 1200 24 2>/dev/null | tail -1 >> $out
MATE_STORE_FORM=1 python tests/soak_vs_oracle.py MATE-4v8-9.yaml 2048 600 24 2>/dev/null | tail -1 | sed 's/^/MATE_STORE_FORM=1 /' >> $out
python tests/soak_vs_oracle.py MATE-4v8-9.yaml 4096 300 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-4v8-9.yaml 1024 300 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-8v8-9.yaml 2048 600 25 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-4v8-0.yaml 2048 600 32 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-Navigation.yaml 2048 600 32 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-4v2-9.yaml 2048 600 32 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-4v4-9.yaml 2048 600 32 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-2v4-0.yaml 2048 400 32 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-8v8-0.yaml 1024 400 32 2>/dev/null | tail -1 >> $out
python tests/soak_vs_oracle.py MATE-1v1-9.yaml 1024 400 32 2>/dev/null | tail -1 >> $out
# ... and the small scenarios with FOUR ENVIRONMENTS PER WAVE forced (MATE_SUBWAVE=1: the soak's batches are below the size from which they are the default)
for w in MATE-2v4-0 MATE-4v2-9 MATE-1v1-9 MATE-4v4-9 MATE-2v2-0; do
  MATE_SUBWAVE=1 python tests/soak_vs_oracle.py $w.yaml 2048 400 32 2>/dev/null | tail -1 | sed 's/^/MATE_SUBWAVE=1 /' >> $out
done
python tests/soak_greedy_vs_oracle.py 2>/dev/null | tail -3 >> $out      # (8v8-9, 4v8-9, Navigation: the script's own list)
cat $out
