#!/bin/bash
# Profiles of one round, run ON THE GPU BOX (gpurun -- 'bash tools/profile_run.sh r02'): kernel trace + stats of the default
# bench command, then separate rocprofv3 --pmc passes (SQ / FETCH_SIZE / WRITE_SIZE: one pass each, MI355X_MICROARCH.md
# "rocprofv3 PMC slots") over one 32 Mb Encoder forward in both arithmetic modes and over one Decoder forward.
# Everything lands in gpurun_out/<tag>/; tools/profile_collect.py turns it into the summaries committed under profiles/.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-configs > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python $ROOT/tools/trace_gaps.py $OUT/bench/bench_kernel_trace.csv 10 > $OUT/bench_gaps.txt 2>&1
python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
for mode in f16x2 bf16; do
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_enc_${mode}_sq -o p -- python $ROOT/tools/prof_encoder.py 32 $mode 1 codes > $OUT/pmc_enc_${mode}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_enc_${mode}_fetch -o p -- python $ROOT/tools/prof_encoder.py 32 $mode 1 codes > $OUT/pmc_enc_${mode}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_enc_${mode}_write -o p -- python $ROOT/tools/prof_encoder.py 32 $mode 1 codes > $OUT/pmc_enc_${mode}_write.log 2>&1
done
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/pmc_dec_sq -o p -- python $ROOT/tools/prof_decoder.py > $OUT/pmc_dec_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_dec_fetch -o p -- python $ROOT/tools/prof_decoder.py > $OUT/pmc_dec_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_dec_write -o p -- python $ROOT/tools/prof_decoder.py > $OUT/pmc_dec_write.log 2>&1
python $ROOT/tools/run_configs.py config3 > $OUT/configs.json 2> $OUT/configs.err
python $ROOT/tools/run_configs.py config5_1024 > $OUT/config5_1024.json 2> $OUT/config5_1024.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_enc_bf16_lds -o p -- python $ROOT/tools/prof_encoder.py 32 bf16 1 codes > $OUT/pmc_enc_bf16_lds.log 2>&1
python $ROOT/tools/time_sv_drivers.py 12 > $OUT/sv_drivers.json 2> $OUT/sv_drivers.err
python $ROOT/tools/range_headroom.py > $OUT/range_headroom.json 2> $OUT/range_headroom.err
# VERDICT r5 #5 (stage 1 off HBM on the P16 path): timing-only ablations of the 64 -> 64 planar conv at n = 32 M, random activations
[ -x $ROOT/tools/microbench_p16 ] && timeout 300 $ROOT/tools/microbench_p16 32000000 r > $OUT/microbench_p16_conv1b.txt 2>&1
( rocm-smi --showpower --showclocks --showtemp > $OUT/rocm_smi_idle.txt 2>&1 ) || true
find $OUT -name "*.csv" | head -40 > $OUT/files.txt
# keep the merge-back under 64 MiB: drop the per-dispatch traces of the counter passes (the collected counters stay)
find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
du -sh $OUT > $OUT/size.txt
