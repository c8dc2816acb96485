#!/bin/bash
# round 4, the committed evidence of the round (profiles/r04_*; $1 = the commit of the kernel sources in this snapshot): upload times,
# the multi-device handle (gather time, host time of a dispatch), then per config the counter passes, the bench lines and the kernel
# trace (tools/profile_round.sh), lane statistics of BOTH leaf modes from the diagnostic build, per-bounce counters of config 1
set -o pipefail
commit=${1:?usage: tools/sessions/r04/evidence.sh <commit of the kernel sources in this snapshot>}
root=$PWD; mkdir -p gpurun_out/ev gpurun_out/prof_r04
python tools/time_upload.py > gpurun_out/ev/upload.log 2>&1; cat gpurun_out/ev/upload.log
timeout -k 10 300 python tools/multi_gather_time.py gpurun_out/ev/multi_gather.json > gpurun_out/ev/multi_gather.out 2> gpurun_out/ev/multi_gather.err || tail -5 gpurun_out/ev/multi_gather.err
for n in 1 2 4 8; do
  timeout -k 10 300 python bench.py --gpus $n --single-process --rehearse --config 4 --steps 2 2> gpurun_out/ev/single$n.err | tail -1 > gpurun_out/ev/single_process_n$n.json || tail -3 gpurun_out/ev/single$n.err
done
python3 - <<'PY'
import json
out = {}
for n in (1, 2, 4, 8):
    try:
        d = json.load(open(f'gpurun_out/ev/single_process_n{n}.json'))
        out[n] = {k: d[k] for k in ('value', 'ms_per_step', 'enqueue_ms_per_step', 'gather_ms', 'rehearsal', 'upload_ms')}
    except Exception as e:
        out[n] = str(e)
json.dump({"note": "bench.py --gpus N --single-process --rehearse --config 4: N contexts on ONE device (loopback copies), the host time of enqueuing a 64-frame step on N contexts from one thread", "runs": out}, open('gpurun_out/prof_r04/r04_multi.json', 'w'), indent=1)
print(json.dumps(out))
PY
for c in 1 3 2 4; do timeout -k 10 800 bash tools/profile_round.sh r04 $c $commit > gpurun_out/prof_r04_cfg$c.log 2>&1 || { tail -5 gpurun_out/prof_r04_cfg$c.log; exit 1; }; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r04/r04_cfg${c}_bench.json')); print('cfg$c', d['value'], d['kernel_ms_rank0'], d['gpu_ms_rank0'], d['roofline']['kernel'], d['roofline']['frac'], d.get('leaves_compare'))"; done
if [ -f wgpu-path-tracing_amd/lib/ab/libptmi_util.so ]; then for c in 1 2 3; do for lv in 2 1; do
  PTMI_OPTS="{\"leaves\": $lv}" PTMI_LIB=$PWD/wgpu-path-tracing_amd/lib/ab/libptmi_util.so timeout -k 10 300 python3 tools/lane_stats.py $c > gpurun_out/prof_r04/r04_cfg${c}_lane_stats_leaves$lv.json 2> gpurun_out/prof_r04/lanes$c$lv.err || exit 1; done; done; fi
python3 bench.py --config 0 > gpurun_out/prof_r04/r04_cfg0_bench.json 2> gpurun_out/prof_r04/r04_cfg0.err; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r04/r04_cfg0_bench.json')); print('cfg0', d['value'], d['kernel_ms_rank0'], d['roofline']['kernel'], d['roofline']['frac'])"
cp gpurun_out/ev/upload.log gpurun_out/prof_r04/r04_upload_times.txt; cp gpurun_out/ev/multi_gather.json gpurun_out/prof_r04/r04_multi_gather.json 2>/dev/null
# per-bounce counters of config 1, one stream, both leaf modes (the account of round 3's session 7)
g1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"; g2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU"
cd /tmp; export TMPDIR=/tmp; cd $root
for lv in 2 1; do
  out=$root/gpurun_out/ev/pb$lv; mkdir -p $out; i=0
  for grp in "$g1" "$g2"; do i=$((i+1)); timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/g$i -- python3 bench.py --no-cpu-baseline --no-leaves-compare --config 1 --overlap 0 --steps 1 --leaves $lv > /dev/null 2> $out/g$i.err || echo "group $i failed"; done
  python3 tools/pmc_per_bounce.py $(find $out -name "*counter_collection.csv") > gpurun_out/prof_r04/r04_cfg1_per_bounce_counters_leaves$lv.json; rm -rf $out
done
