#!/bin/bash
# round-end measurement set (run on the GPU box through gpurun, in three calls: `final_measure.sh a`, `... b`, `... c`;
# each stays under gpurun's 20-minute limit)
R=$GRAFT_REPO_ROOT
O=${KWY_MEASURE_OUT:-$R/gpurun_out/final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
set -e
PART=${1:-a}
SERIAL="python $R/bench.py --driver serial --batch 16 --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline"
if [ "$PART" = a ]; then
  KWY_KAT_DUMP=$O/kat_envelopes timeout -k 10 900 python -m pytest $R/tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
  python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
  python $R/bench.py --batch 16 --waves 1 --no-cpu-baseline > $O/bench_b16.json 2>/dev/null
  python $R/bench.py --batch 64 --steps 10 --no-cpu-baseline > $O/bench_b64.json 2>/dev/null
  python $R/bench.py --batch 1 --waves 1 --steps 20 --no-cpu-baseline --no-variants > $O/bench_b1.json 2>/dev/null
  python $R/bench.py --driver serial --no-cpu-baseline --no-variants > $O/bench_serial.json 2>/dev/null
  python $R/bench.py --driver streams --no-cpu-baseline > $O/bench_streams.json 2>/dev/null
  python $R/bench.py --workload utterance --batch 1 > $O/bench_utt_b1.json 2>/dev/null
  python $R/bench.py --workload utterance --utterances 256 --steps 3 --warmup 1 > $O/bench_config4.json 2> $O/bench_config4.err
  # round 5: the same step from waveforms to 16-bit samples (DIO + StoneMask and the post-step inside it)
  python $R/bench.py --workload wav --no-variants --config4 off > $O/bench_wav.json 2> $O/bench_wav.err
  # the serial chain's streams at a higher stream priority (round 4's review, item 8): recorded, not the default
  python $R/bench.py --chain-priority on --no-variants --config4 off --no-cpu-baseline > $O/bench_chain_priority.json 2>/dev/null
  echo done > $O/DONE_A
elif [ "$PART" = b ]; then
  python $R/bench_fit.py > $O/bench_fit.json 2>/dev/null
  python $R/bench_corpus.py > $O/bench_corpus.json 2> $O/bench_corpus.err
  python $R/bench_corpus.py --profile-fit > $O/bench_corpus_profiled.json 2>/dev/null
  python $R/bench_corpus.py --em-iters 10 > $O/bench_corpus_em10.json 2>/dev/null
  python $R/bench_corpus.py --driver streams --distinct 8 --em-iters 10 > $O/bench_corpus_streams.json 2>/dev/null
  # the N-rank paths rehearsed on this one GPU: two ranks started by the benches' OWN --gpus flag (bench_launch.py),
  # gloo for the collectives
  python $R/bench_fit.py --gpus 2 --backend gloo --frames 200000 > $O/bench_fit_2rank_gloo.json 2> $O/bench_fit_2rank_gloo.err
  python $R/bench_corpus.py --gpus 2 --backend gloo --pairs 64 --seconds 2 --em-iters 10 > $O/bench_corpus_2rank_gloo.json 2> $O/bench_corpus_2rank_gloo.err
  python $R/bench_corpus.py --pairs 64 --seconds 2 --em-iters 10 > $O/bench_corpus_1rank_64.json 2>/dev/null
  python $R/bench.py --gpus 2 --backend gloo --batch 8 --steps 5 --config4-utterances 32 --no-cpu-baseline --no-variants > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
  # RCCL itself: a one-rank 'nccl' group on the one GPU
  python $R/bench_fit.py --backend nccl --force-group > $O/bench_fit_1rank_nccl.json 2> $O/bench_fit_1rank_nccl.err
  echo done > $O/DONE_B
else
  KWY_MEASURE_OUT=$O/prof_step bash $R/tools/prof_step.sh
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $SERIAL > $O/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $SERIAL > $O/pmc_write.log 2>&1
  KWY_MEASURE_OUT=$O/pmc_sq bash $R/tools/pmc_sq.sh
  bash $R/tools/prof_fit.sh
  # round 5: per-kernel durations of the wav-in / pcm-out step (k_dio_*, k_stonemask, k_fin_*) and of the batch path's
  # differential outputs (k_mlsa_filter, k_mc2b)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/wav_serial -- python $R/bench.py --workload wav --driver serial --no-variants --no-cpu-baseline --config4 off > $O/wav_serial.log 2>&1
  cp "$(ls -t $O/wav_serial/*/*kernel_stats.csv | head -1)" $O/wav_serial_kernel_stats.csv
  rm -f $O/wav_serial/*/*kernel_trace.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/diff_batch -- python $R/tools/diff_batch_run.py > $O/diff_batch.log 2>&1
  cp "$(ls -t $O/diff_batch/*/*kernel_stats.csv | head -1)" $O/diff_batch_kernel_stats.csv
  rm -f $O/diff_batch/*/*kernel_trace.csv
  echo done > $O/DONE_C
fi
