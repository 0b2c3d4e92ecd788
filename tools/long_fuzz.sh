#!/bin/bash
# The randomised cross-checks at several times their usual length (soaks, fuzzers), last line of each to stdout: about three
# minutes on the GPU box; run from the repository root (profiles/r06_long_fuzz.txt holds the round's run).
python tools/headline_soak.py 96 > gpurun_out/long_headline_soak.log 2>&1; tail -1 gpurun_out/long_headline_soak.log
python tools/group_soak.py 60 > gpurun_out/long_group_soak.log 2>&1; tail -1 gpurun_out/long_group_soak.log
python tools/ws_fuzz.py 2000 23 > gpurun_out/long_ws_fuzz.log 2>&1; tail -1 gpurun_out/long_ws_fuzz.log
python tools/mg_fuzz.py 80 7 > gpurun_out/long_mg_fuzz.log 2>&1; tail -1 gpurun_out/long_mg_fuzz.log
python tools/carry_fuzz.py $(seq 12 35) > gpurun_out/long_carry_fuzz.log 2>&1; tail -1 gpurun_out/long_carry_fuzz.log
python tools/covariance_fuzz.py 200 5 > gpurun_out/long_covariance_fuzz.log 2>&1; tail -1 gpurun_out/long_covariance_fuzz.log
python tools/on_chip_fuzz.py 600 7 > gpurun_out/long_on_chip_fuzz.log 2>&1; tail -1 gpurun_out/long_on_chip_fuzz.log
