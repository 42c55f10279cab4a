#!/usr/bin/env node
/*
 * time_reference.js -- times the REFERENCE's own CPU path (BUILD CONTAINER ONLY: needs /root/reference).
 *
 * Loads the unmodified worklet bundle from its mounted location exactly as tests/golden/gen_golden.js does (two size literals
 * substituted at load time, AudioWorkletProcessor / registerProcessor shimmed; the bundle text is never copied), drives
 * process() single-threaded with the seeded "tonal" signal of SURVEY section 4 and prints one JSON line per configuration:
 *   {"config": "C2", "shape": "1024/256", "nch": 1, "pitch": 1.5, "frames": ..., "seconds": ..., "frames_per_s": ...}
 * tools/time_reference.py runs this (pinned to one core), times the C port (oracle/) on the same core in the same run and
 * writes profiles/cpu_reference_ratio.json, which bench.py uses to quote an estimate of the reference's rate on the GPU
 * box's host next to the measured port (the reference itself cannot travel there).
 */
'use strict';
const fs = require('fs');
const vm = require('vm');
const REF = process.env.PHAZE_REFERENCE_BUNDLE || '/root/reference/www/phase-vocoder.js';

function loadProcessorClass(fftSize, hop) {
  let src = fs.readFileSync(REF, 'utf8');
  const a = 'const BUFFERED_BLOCK_SIZE = 2048;', b = 'const WEBAUDIO_BLOCK_SIZE = 128;';
  if (!src.includes(a) || !src.includes(b)) throw new Error('size literals not found in bundle');
  src = src.replace(a, 'const BUFFERED_BLOCK_SIZE = ' + fftSize + ';').replace(b, 'const WEBAUDIO_BLOCK_SIZE = ' + hop + ';');
  let registered = null;
  const sandbox = { AudioWorkletProcessor: class { constructor(o) {} }, registerProcessor: (name, cls) => { registered = { name, cls }; }, console };
  vm.runInNewContext(src, sandbox, { filename: 'phase-vocoder.bundle.js' });
  if (!registered) throw new Error('processor not registered');
  return registered.cls;
}
function lcgNoise(seed, n, amp) {
  const x = new Float32Array(n);
  let s = seed >>> 0;
  for (let i = 0; i < n; i++) { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; x[i] = ((s >>> 8) - 8388608) / 8388608 * amp; }
  return x;
}
function tri(i, P) { return 4 * Math.abs((i % P) / P - 0.5) - 1; }
function tonal(ch, n) {
  const nz = lcgNoise(2000 + ch, n, 1 / 64), x = new Float32Array(n);
  for (let i = 0; i < n; i++) x[i] = 0.25 * tri(i, 109) + 0.125 * tri(i, 31) + 0.0625 * tri(i, 7) + nz[i];
  return x;
}
const CONFIGS = [   // BASELINE.json configs[0..4] + the reference's native shape
  { config: 'C1', fft: 1024, hop: 256, nch: 1, pitch: () => 1.0 },
  { config: 'C2', fft: 1024, hop: 256, nch: 1, pitch: () => 1.5 },
  { config: 'C3', fft: 2048, hop: 512, nch: 2, pitch: () => Math.fround(0.8) },
  { config: 'C4', fft: 4096, hop: 1024, nch: 8, pitch: () => 1.25 },
  { config: 'C5', fft: 8192, hop: 2048, nch: 8, pitch: (m) => Math.fround(0.5 + 1.5 * (m % 64) / 63) },
  { config: 'native', fft: 2048, hop: 128, nch: 1, pitch: () => 1.5 },
];
const seconds = Number(process.argv[2] || 3);
for (const c of CONFIGS) {
  const Cls = loadProcessorClass(c.fft, c.hop);
  const proc = new Cls({ numberOfInputs: 1, numberOfOutputs: 1 });
  const L = 256, sig = [];
  for (let ch = 0; ch < c.nch; ch++) sig.push(tonal(ch, L * c.hop));
  const inputs = [[]], outputs = [[]];
  for (let ch = 0; ch < c.nch; ch++) { inputs[0].push(new Float32Array(c.hop)); outputs[0].push(new Float32Array(c.hop)); }
  const pf = new Float32Array(1);
  const call = (m) => {
    for (let ch = 0; ch < c.nch; ch++) inputs[0][ch].set(sig[ch].subarray((m % L) * c.hop, (m % L + 1) * c.hop));
    pf[0] = c.pitch(m);
    proc.process(inputs, outputs, { pitchFactor: pf });
  };
  let m = 0;
  for (; m < 300; m++) call(m);                                  // JIT warm-up
  const t0 = process.hrtime.bigint();
  let calls = 0, dt = 0;
  do { for (let i = 0; i < 50; i++, m++, calls++) call(m); dt = Number(process.hrtime.bigint() - t0) * 1e-9; } while (dt < seconds);
  console.log(JSON.stringify({ config: c.config, shape: c.fft + '/' + c.hop, nch: c.nch, pitch: c.config === 'C5' ? 'sweep 0.5->2.0' : c.pitch(0),
                               frames: calls * c.nch, seconds: dt, frames_per_s: calls * c.nch / dt }));
}
