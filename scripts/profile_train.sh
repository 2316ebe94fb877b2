set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_train1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python scripts/bench_train.py --epochs 40 --cpu-epochs 1 > $OUT/trace.log 2>&1
python - <<'PY'
import sys; sys.path.insert(0,'scripts')
import glob, summarise_prof as S
db = glob.glob('gpurun_out/prof_train1/trace/**/*.db', recursive=True)
print(db)
open('gpurun_out/prof_train1_summary.txt','w').write(S.kernel_stats(db[0]))
PY
find $OUT -name "*.db" -delete
tail -2 $OUT/trace.log
python scripts/bench_configs.py 2>&1 | tail -2
