# round 6, session j: the CPU walker by thread count on the GPU host (scan_baseline_parallel; the crew's wake-up cost shows in
# the small files)
for t in 1 2 4 8 16; do ZJ_PAR_DEBUG=1 timeout 300 python tools/walker_bench.py --pinned --no-pillow --threads $t --reps 9 2>&1 | grep -E "^(test-|pillow-)|scan_baseline_parallel|^  chunk [013]:" | grep -v "^  part" | tail -7 | cut -c1-220 | sed "s/^/threads $t: /"; done
