# TIMING ONLY (wrong tokens by construction): what a B = 16 matrix-core launch spends INSIDE the kernel, by knocking parts of it out
# (dbg_skip bits 16-19 (= the old g16_slab bits 4-7, value << 12): 16 no activation-plane loads, 32 no weight-fragment loads (one hot KiB instead), 64 no MFMAs, 128 no plane stores)
O=gpurun_out; mkdir -p $O
# needs the variant build: python -c "from csm_hf_amd import build; build.build_library(defines=('CSM_G16_KO',), out='csm-hf_amd/libcsm_hip_g16ko.so')"
export CSM_HIP_LIB=$PWD/csm-hf_amd/libcsm_hip_g16ko.so
run() { timeout 200 python bench.py --no-cpu-baseline --config4 0 --lean --batch 16 --steps 100 "$@" 2>>$O/g16ko.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-40s ms/step %.4f' % (' '.join(sys.argv[1:]), d['ms_per_step']))
" "$@"; }
{ run; for v in 16 32 64 48 112; do run --opt dbg_skip=$((v << 12)); done; run; unset CSM_HIP_LIB; run; } > $O/g16_inkernel.txt 2>&1
cat $O/g16_inkernel.txt
