cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "wave2k or fuzz or golden or state" 2>&1 | tail -2
run() { PHAZE_LIB=$1 python bench.py --allow-lib-override --no-extras --no-cpu-baseline --steps 30 --warmup 8 --fft 2048 --hop $2 --channels 2 --hops 262144 --pitch $3 | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$4 2048/$2 pf $3', 'ms %.3f'%j['roofline']['kernel_ms'], 'frac %.4f'%j['roofline']['frac'], j['parity_rms_vs_oracle'])"; }
B=$PWD/build/exp/libphaze_base.so; N=$PWD/phaze_amd/lib/libphaze_amd.so
for rep in 1 2; do for a in "512 0.8" "128 1.0" "512 1.5"; do set -- $a; run $B $1 $2 base; run $N $1 $2 new; done; done
