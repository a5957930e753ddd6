#!/bin/bash
# One-GPU shares of the other BASELINE configurations (bench lines for profiles/; not the headline metric).
set -e
run() { timeout -k 10 400 python bench.py --no_cpu_baseline --event_steps 0 "$@" | tail -1; }
run --hand allegro --n_objects 8 --batch_size 256 --steps 200 --warmup 24
run --hand shadow_hand --n_objects 8 --batch_size 512 --n_contact 16 --steps 100 --warmup 16
run --hand robotiq3 --n_objects 8 --batch_size 1024 --n_cone_vecs 8 --steps 40 --warmup 8
run --hand robotiq3 --n_objects 32 --batch_size 1024 --n_cone_vecs 8 --steps 24 --warmup 8
run --hand allegro --n_objects 32 --batch_size 1024 --n_cone_vecs 8 --steps 24 --warmup 8
