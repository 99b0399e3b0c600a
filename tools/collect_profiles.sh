#!/bin/bash
# Copy the summaries of tools/final_measure.sh (merged back into gpurun_out/final by gpurun) into profiles/<round>_*:
#   tools/collect_profiles.sh [round, default r3] [source dir, default gpurun_out/final]
set -e
cd "$(dirname "$0")/.."
RND=${1:-r3}
S=${2:-gpurun_out/final}
P=profiles
newest() { ls -t $1 2>/dev/null | head -1; }
cp $S/bench_default.json $P/${RND}_pair_b32_bench.json
cp $S/bench_b1.json $P/${RND}_pair_b1_bench.json
cp $S/bench_utt_b1.json $P/${RND}_utterance_b1_bench.json
cp $S/bench_fit.json $P/${RND}_fit_bench.json
cp $S/bench_corpus.json $P/${RND}_corpus_bench.json
for f in config4 corpus_ref_em corpus_hostpads fit_2rank_gloo corpus_2rank_gloo corpus_1rank_64 2rank_gloo; do
  [ -s $S/bench_$f.json ] && tail -n 1 $S/bench_$f.json > $P/${RND}_${f}_bench.json   # (gloo prints a line of its own first)
done
cp "$(newest "$S/stats_pair_b32/*/*kernel_stats.csv")" $P/${RND}_pair_b32_kernel_stats.csv
cp "$(newest "$S/stats_pair_b1/*/*kernel_stats.csv")" $P/${RND}_pair_b1_kernel_stats.csv
[ -d $S/stats_pair_b1_single ] && cp "$(newest "$S/stats_pair_b1_single/*/*kernel_stats.csv")" $P/${RND}_pair_b1_single_stream_kernel_stats.csv
F=$(newest "$S/pmc_fetch/*/*counter_collection.csv"); W=$(newest "$S/pmc_write/*/*counter_collection.csv")
cp "$F" $P/${RND}_pmc_fetch_counter_collection.csv
cp "$W" $P/${RND}_pmc_write_counter_collection.csv
python $P/make_pmc_traffic.py "$F" "$W" $RND > /dev/null
[ -f $S/kat_envelopes.hip.json ] && cp $S/kat_envelopes.hip.json $P/${RND}_kat_envelopes_hip.json
if [ -d $S/pmc_sq ]; then
  cp "$(newest "$S/pmc_sq/pass1/*/*counter_collection.csv")" $P/${RND}_pmc_sq_pass1_counter_collection.csv
  cp "$(newest "$S/pmc_sq/pass2/*/*counter_collection.csv")" $P/${RND}_pmc_sq_pass2_counter_collection.csv
  cp $S/pmc_sq/summary.txt $P/${RND}_pmc_sq_summary.json
fi
[ -f gpurun_out/prof_fit/kernel_stats.csv ] && cp gpurun_out/prof_fit/kernel_stats.csv $P/${RND}_fit_kernel_stats.csv
ls -la $P | grep ${RND}_
