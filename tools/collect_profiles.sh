#!/bin/bash
# Copy the summaries of tools/final_measure.sh (merged back into gpurun_out/final by gpurun) into profiles/<round>_*:
#   tools/collect_profiles.sh [round, default r4] [source dir, default gpurun_out/final]
set -e
cd "$(dirname "$0")/.."
RND=${1:-r5}
S=${2:-gpurun_out/final}
P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
keep() { [ -s "$1" ] && tail -n 1 "$1" > "$2" || true; }     # (gloo prints a line of its own first: the JSON is the last line)
keep $S/bench_default.json $P/${RND}_pair_b32_bench.json
keep $S/bench_b16.json $P/${RND}_pair_b16_bench.json
keep $S/bench_b64.json $P/${RND}_pair_b64_bench.json
keep $S/bench_b1.json $P/${RND}_pair_b1_bench.json
keep $S/bench_serial.json $P/${RND}_pair_serial_bench.json
keep $S/bench_streams.json $P/${RND}_pair_streams_bench.json
keep $S/bench_utt_b1.json $P/${RND}_utterance_b1_bench.json
keep $S/bench_config4.json $P/${RND}_config4_bench.json
keep $S/bench_wav.json $P/${RND}_wav_in_pcm_out_bench.json
keep $S/bench_chain_priority.json $P/${RND}_chain_priority_bench.json
keep $S/bench_corpus_profiled.json $P/${RND}_corpus_profiled_bench.json
keep $S/bench_fit.json $P/${RND}_fit_bench.json
keep $S/bench_fit_1rank_nccl.json $P/${RND}_fit_1rank_nccl_bench.json
keep $S/bench_corpus.json $P/${RND}_corpus_bench.json
keep $S/bench_corpus_em10.json $P/${RND}_corpus_em10_bench.json
keep $S/bench_corpus_streams.json $P/${RND}_corpus_streams_bench.json
for f in fit_2rank_gloo corpus_2rank_gloo corpus_1rank_64 2rank_gloo; do keep $S/bench_$f.json $P/${RND}_${f}_bench.json; done
if [ -d $S/prof_step ]; then
  cp $S/prof_step/step_busy.json $P/${RND}_step_busy.json
  cp $S/prof_step/serial_kernel_stats.csv $P/${RND}_pair_serial_kernel_stats.csv
fi
F=$(newest "$S/pmc_fetch/*/*counter_collection.csv"); W=$(newest "$S/pmc_write/*/*counter_collection.csv")
if [ -n "$F" ] && [ -n "$W" ]; then
  cp "$F" $P/${RND}_pmc_fetch_counter_collection.csv
  cp "$W" $P/${RND}_pmc_write_counter_collection.csv
  python $P/make_pmc_traffic.py "$F" "$W" $RND 33616 \
    "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace (two separate passes) -- python bench.py --driver serial --batch 16 --steps 2 --warmup 1 --no-graph --no-variants --no-cpu-baseline" > /dev/null
fi
[ -f $S/wav_serial_kernel_stats.csv ] && cp $S/wav_serial_kernel_stats.csv $P/${RND}_wav_serial_kernel_stats.csv
[ -f $S/diff_batch_kernel_stats.csv ] && cp $S/diff_batch_kernel_stats.csv $P/${RND}_diff_batch_kernel_stats.csv
[ -f $S/diff_batch.log ] && grep "^{" $S/diff_batch.log | tail -n 1 > $P/${RND}_diff_batch_bench.json
[ -f $S/kat_envelopes.hip.json ] && cp $S/kat_envelopes.hip.json $P/${RND}_kat_envelopes_hip.json
if [ -d $S/pmc_sq ]; then
  cp "$(newest "$S/pmc_sq/pass1/*/*counter_collection.csv")" $P/${RND}_pmc_sq_pass1_counter_collection.csv
  cp "$(newest "$S/pmc_sq/pass2/*/*counter_collection.csv")" $P/${RND}_pmc_sq_pass2_counter_collection.csv
  cp $S/pmc_sq/summary.txt $P/${RND}_pmc_sq_summary.json
fi
[ -f gpurun_out/prof_fit/kernel_stats.csv ] && cp gpurun_out/prof_fit/kernel_stats.csv $P/${RND}_fit_kernel_stats.csv
ls -la $P | grep ${RND}_
