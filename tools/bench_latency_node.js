#!/usr/bin/env node
/*
 * bench_latency_node.js -- per-call latency histogram of the STREAMING form through the actual product boundary:
 * PhaseVocoderProcessor.process(inputs, outputs, {pitchFactor}) of phaze_amd/node/phase-vocoder.js -> N-API addon -> C ABI -> HIP kernel.
 * Caller shape: /root/reference/src/ola-processor.js:159-171 (one call per render quantum of hopSize samples).
 * Prints one JSON line per configuration (run on the GPU box; results are kept under profiles/).
 *   node tools/bench_latency_node.js [calls] [--inputs K]
 * --inputs K adds the native shape with numberOfInputs = K (K independent processors per quantum, phase-vocoder.js:49-50): the host launches
 * all K handles before it waits for the first (processBegin / processEnd); `sequential_p50` is the same quantum driven one handle after the
 * other (launch + wait each), what the addon did before round 3.
 */
"use strict";
const path = require("path");
const { PhaseVocoderProcessor } = require(path.join(__dirname, "..", "phaze_amd", "node", "phase-vocoder.js"));
const argv = process.argv.slice(2);
const ri = argv.indexOf("--resident");
const resident = ri >= 0;
if (ri >= 0) argv.splice(ri, 1);
const ki = argv.indexOf("--inputs");
const extraInputs = ki >= 0 ? Number(argv[ki + 1]) : 0;
if (ki >= 0) argv.splice(ki, 2);
const calls = Number(argv[0] || 3000);

function lcg(seed, n, amp) { const x = new Float32Array(n); let s = seed >>> 0; for (let i = 0; i < n; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; x[i] = ((s >>> 8) - 8388608) / 8388608 * amp; } return x; }

function run(cfg) {
  const { fft, hop, nch, fs, sweep, label } = cfg;
  const nin = cfg.inputs || 1;
  const opts = { numberOfInputs: nin, numberOfOutputs: nin };
  if (!(fft === 2048 && hop === 128)) opts.processorOptions = { fftSize: fft, hopSize: hop };     // native shape: the reference's own defaults
  if (cfg.flags) opts.processorOptions = Object.assign(opts.processorOptions || {}, { flags: cfg.flags });
  const proc = new PhaseVocoderProcessor(opts);
  const L = 64, sig = [];
  for (let c = 0; c < nch; c++) { const nz = lcg(2000 + c, L * hop, 1 / 64), x = new Float32Array(L * hop); for (let i = 0; i < x.length; i++) x[i] = 0.25 * Math.sin(i * 0.031 * (c + 1)) + nz[i]; sig.push(x); }
  const inputs = [], outputs = [];
  for (let i = 0; i < nin; i++) { inputs.push([]); outputs.push([]); for (let c = 0; c < nch; c++) { inputs[i].push(null); outputs[i].push(new Float32Array(hop)); } }
  const pf = new Float32Array(1);
  const lat = new Float64Array(calls);
  for (let m = 0; m < calls + 50; m++) {
    for (let i = 0; i < nin; i++) for (let c = 0; c < nch; c++) inputs[i][c] = sig[c].subarray(((m + 7 * i) % L) * hop, ((m + 7 * i) % L + 1) * hop);
    pf[0] = sweep ? 0.5 + 1.5 * (m % 64) / 63 : 1.5;
    const t0 = process.hrtime.bigint();
    proc.process(inputs, outputs, { pitchFactor: pf });
    const t1 = process.hrtime.bigint();
    if (m >= 50) lat[m - 50] = Number(t1 - t0) * 1e-3;
  }
  const info = proc.info();
  let seqP50 = null;
  if (nin > 1) {                                                   // the same quantum, one handle after the other (launch + wait each)
    const native = require(path.join(__dirname, "..", "phaze_amd", "node", "phase-vocoder.js")).native;
    const seq = new Float64Array(calls);
    for (let m = 0; m < calls + 50; m++) {
      for (let i = 0; i < nin; i++) for (let c = 0; c < nch; c++) inputs[i][c] = sig[c].subarray(((m + 7 * i) % L) * hop, ((m + 7 * i) % L + 1) * hop);
      const t0 = process.hrtime.bigint();
      for (let i = 0; i < nin; i++) native.process(proc._handles[i], inputs[i], outputs[i], 1.5);
      const t1 = process.hrtime.bigint();
      if (m >= 50) seq[m - 50] = Number(t1 - t0) * 1e-3;
    }
    seqP50 = Array.from(seq).sort((x, y) => x - y)[Math.floor(calls / 2)];
  }
  proc.close();
  const a = Array.from(lat).sort((x, y) => x - y), q = (p) => a[Math.min(a.length - 1, Math.floor(p / 100 * a.length))];
  const edges = [0, 25, 50, 75, 100, 150, 200, 300, 500, 1000], hist = new Array(edges.length).fill(0);
  for (const v of a) { let b = edges.length - 1; while (b > 0 && v < edges[b]) b--; hist[b]++; }
  const mean = a.reduce((s, v) => s + v, 0) / a.length, budget = hop / fs * 1e6;
  return { metric: "stream_call_latency_us", boundary: "PhaseVocoderProcessor.process (Node host -> N-API -> C ABI)", config: { workload: label, calls, kernel: info.kernelName },
           p50: q(50), p90: q(90), p99: q(99), max: a[a.length - 1], mean, realtime_budget_us: budget, budget_over_p99: budget / q(99),
           histogram_us_edges: edges.concat(["inf"]).slice(0, edges.length).map(String), histogram_counts: hist, frames_per_s_streaming: nin * nch / (mean * 1e-6), inputs: nin, sequential_p50: seqP50, flags: cfg.flags | 0, node: process.version };
}
for (const cfg of [
  { fft: 8192, hop: 2048, nch: 8, fs: 96000, sweep: true, label: "BASELINE configs[4]: 8-ch 96 kHz FFT=8192 hop=2048, pitchFactor swept 0.5->2.0 per hop" },
  { fft: 2048, hop: 128, nch: 2, fs: 48000, sweep: false, label: "reference native shape: stereo 48 kHz FFT=2048 hop=128 (processor defaults), pitchFactor 1.5" },
  { fft: 1024, hop: 256, nch: 1, fs: 48000, sweep: false, label: "BASELINE configs[1] shape, streaming: mono 48 kHz FFT=1024 hop=256, pitchFactor 1.5" },
].concat(resident ? [{ fft: 1024, hop: 256, nch: 1, fs: 48000, sweep: false, flags: 32, label: "BASELINE configs[1] shape, streaming on the RESIDENT kernel (PV_FLAG_PERSISTENT_STREAM): mono 48 kHz FFT=1024 hop=256, pitchFactor 1.5" },
  { fft: 1024, hop: 256, nch: 2, fs: 48000, sweep: true, flags: 32, label: "stereo 48 kHz FFT=1024 hop=256 on the resident kernel, pitchFactor swept 0.5->2.0" },
  { fft: 2048, hop: 128, nch: 2, fs: 48000, sweep: false, flags: 32, label: "reference native shape on the RESIDENT kernel: stereo 48 kHz FFT=2048 hop=128, pitchFactor 1.5" },
  { fft: 8192, hop: 2048, nch: 8, fs: 96000, sweep: true, flags: 32, label: "BASELINE configs[4] on the RESIDENT kernel: 8-ch 96 kHz FFT=8192 hop=2048, pitchFactor swept 0.5->2.0 per hop" }] : []).concat(extraInputs > 1 ? [{ fft: 2048, hop: 128, nch: 2, fs: 48000, sweep: false, inputs: extraInputs,
    label: `reference native shape with numberOfInputs = ${extraInputs}: ${extraInputs} x stereo 48 kHz FFT=2048 hop=128, pitchFactor 1.5` }] : [])) console.log(JSON.stringify(run(cfg)));
