#!/usr/bin/env node
/*
 * bench_node_sharded.js -- many independent streams on the GPUs of one node, driven from ONE Node.js process through the product boundary
 * (phaze_amd/node/sharded.js -> N-API processBatchAsync -> pv_process_batch on libuv worker threads).  Host-buffer form: the figure includes
 * the PCIe copies each way (a Node host owns host memory); the HBM-resident kernel rate is bench.py's.  Like bench.py it measures what
 * exists and says so: `requested_gpus` vs `replicas_measured`, nothing is extrapolated.  Default: the host writes its streams IN PLACE into the
 * shards' pinned buffers (inputView / outputView: the timed region is processInPlace, i.e. DMA + kernels, what a host that produces its samples
 * there pays); --copy times the round-3 form (samples in the host's own arrays: one JS copy in and out per channel on top).
 *   node tools/bench_node_sharded.js [--gpus N] [--streams S] [--channels C] [--fft N] [--hop H] [--hops T] [--steps K] [--pitch F] [--copy]
 * With --gpus G > devices present the G shards share the devices; `shard_concurrency` then reports how much faster G shards in flight together
 * finish than G times one shard alone (>= ~1: the libuv pool really runs them side by side; UV_THREADPOOL_SIZE is raised before the pool starts).
 */
"use strict";
const path = require("path");
const arg = (name, dflt) => { const i = process.argv.indexOf("--" + name); return i >= 0 ? Number(process.argv[i + 1]) : dflt; };
const gpus = arg("gpus", 0), fft = arg("fft", 4096), hop = arg("hop", 1024), cps = arg("channels", 8), streams = arg("streams", 32);
const T = arg("hops", 64), steps = arg("steps", 10), pitchF = arg("pitch", 1.25);
if (gpus > 4) process.env.UV_THREADPOOL_SIZE = String(gpus);            // before anything touches the libuv thread pool
const { ShardedPhaseVocoder } = require(path.join(__dirname, "..", "phaze_amd", "node", "sharded.js"));

function lcg(seed, n, amp) { const x = new Float32Array(n); let s = seed >>> 0; for (let i = 0; i < n; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; x[i] = ((s >>> 8) - 8388608) / 8388608 * amp; } return x; }

(async () => {
  const pv = new ShardedPhaseVocoder({ fftSize: fft, hopSize: hop, channelsPerStream: cps, streams, maxHops: T, gpus: gpus || undefined });
  const n = T * hop;
  const inputs = [], outputs = [], pitch = [];
  for (let s = 0; s < streams; s++) {
    const ins = [], outs = [];
    for (let c = 0; c < cps; c++) { const nz = lcg(2000 + c + 100000 * s, n, 1 / 64), x = new Float32Array(n); for (let i = 0; i < n; i++) x[i] = 0.25 * Math.sin(i * 0.031 * (c + 1 + s % 5)) + nz[i]; ins.push(x); outs.push(new Float32Array(n)); }
    inputs.push(ins); outputs.push(outs); pitch.push(new Float32Array(T).fill(pitchF));
  }
  const copyForm = process.argv.includes("--copy");
  for (let s = 0; s < streams; s++) { for (let c = 0; c < cps; c++) pv.inputView(s, c).set(inputs[s][c]); pv.pitchView(s).set(pitch[s]); }
  const run = () => copyForm ? pv.processBatch(inputs, outputs, pitch, T) : pv.processInPlace(T);
  await run();                                                           // warm-up (first launches, stream / event creation)
  const t0 = process.hrtime.bigint();
  for (let k = 0; k < steps; k++) await run();
  const dt = Number(process.hrtime.bigint() - t0) * 1e-9;
  const o0 = copyForm ? outputs[0][0] : pv.outputView(0, 0);
  let e = 0; for (const o of o0) e += o * o;
  const info = pv.info();
  console.log(JSON.stringify({
    metric: "stft_frames_per_sec_node_sharded", value: steps * streams * cps * T / dt, unit: "frames/s", frames_per_s: steps * streams * cps * T / dt,
    n_gpus: pv.replicasMeasured, requested_gpus: pv.requestedGpus, replicas_measured: pv.replicasMeasured, shards: pv.shards, devices_present: pv.devicesPresent,
    steps, ms_per_step: dt / steps * 1e3,
    form: (copyForm ? "host arrays copied into / out of the shards' pinned buffers in JS (ShardedPhaseVocoder.processBatch)" : "streams written in place into the shards' pinned buffers (ShardedPhaseVocoder.processInPlace)")
          + "; PCIe copies included, pipelined per shard; all shards in flight before the first wait",
    bytes_each_way_per_step: streams * cps * T * hop * 4, gbytes_per_s_each_way: steps * streams * cps * T * hop * 4 / dt / 1e9,
    config: { workload: `${streams} streams x ${cps} ch, FFT=${fft} hop=${hop}, ${T} hops per step, pitchFactor ${pitchF}; stream s -> shard s mod ${pv.shards}`,
              kernel: info[0].kernelName, device: info[0].deviceName, uv_threadpool_size: pv.threadPoolSize },
    shards_in_flight_together: pv.maxConcurrentShards(),
    output_rms_stream0: Math.sqrt(e / o0.length), node: process.version }));
  pv.close();
})().catch((e) => { console.error(e); process.exit(1); });
