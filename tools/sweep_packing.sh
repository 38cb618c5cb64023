# knobs that change a kernel family's CU footprint (LDS / registers per workgroup), measured in the 8-stream pipeline where the
# families share the CUs; alternating repeats.   usage: bash tools/sweep_packing.sh "ENV=val ..." "ENV=val ..." ...
run() {
  env $1 timeout 200 python bench.py --no-cpu-baseline --no-h2d-pass --steps 24 --warmup 4 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.readline()); k=l['kernels']
print('%-40s %.1f fps  %.2f ms/step | solver %.2f conv %.2f pointwise %.2f' % ('$1', l['value'], l['ms_per_step'], k['solve_kernel']['ms_per_step'], k['conv2d_kernel']['ms_per_step'], k['pointwise_gemm_kernel(+point_head)']['ms_per_step']))"
}
for rep in 1 2; do for v in "$@"; do run "$v"; done; done
