#!/bin/bash
REPS=3 STEPS=3 bash scripts/ab_bench.sh cornell 2>&1 | tail -5
REPS=1 STEPS=2 bash scripts/ab_bench.sh blob room 2>&1 | tail -9
