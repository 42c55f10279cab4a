#!/usr/bin/env node
/*
 * gen_golden.js -- golden-vector generator (TEST INFRASTRUCTURE, build-container only).
 *
 * Drives the UNMODIFIED reference worklet bundle, read from its mounted location
 * (/root/reference/www/phase-vocoder.js -- never copied into this repo), with the seeded synthetic
 * inputs described in cases.json, and writes the reference's outputs (and a few intermediate dumps)
 * as raw little-endian fixtures next to this script, plus manifest.json (sizes, sha256, rms).
 *
 * The only edits applied to the bundle text, at load time, are the two size literals
 *   `const BUFFERED_BLOCK_SIZE = 2048;`  (src/phase-vocoder.js:6)
 *   `const WEBAUDIO_BLOCK_SIZE = 128;`   (src/ola-processor.js:3)
 * because none of the BASELINE configs is the reference's native 2048/128. Host globals
 * AudioWorkletProcessor / registerProcessor are shimmed (they are browser globals, src/phase-vocoder.js:176).
 *
 * Usage (only where /root/reference exists):   node tests/golden/gen_golden.js
 * The GPU box never runs this; it only reads the committed fixtures.
 */
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const crypto = require('crypto');

const REF = process.env.PHAZE_REFERENCE_BUNDLE || '/root/reference/www/phase-vocoder.js';
const HERE = __dirname;

function loadProcessorClass(fftSize, hop) {
  let src = fs.readFileSync(REF, 'utf8');
  const a = 'const BUFFERED_BLOCK_SIZE = 2048;';
  const b = 'const WEBAUDIO_BLOCK_SIZE = 128;';
  if (!src.includes(a) || !src.includes(b)) throw new Error('size literals not found in bundle');
  src = src.replace(a, 'const BUFFERED_BLOCK_SIZE = ' + fftSize + ';');
  src = src.replace(b, 'const WEBAUDIO_BLOCK_SIZE = ' + hop + ';');
  let registered = null;
  const sandbox = {
    AudioWorkletProcessor: class { constructor(o) {} },
    registerProcessor: (name, cls) => { registered = { name, cls }; },
    console: console,
  };
  vm.runInNewContext(src, sandbox, { filename: 'phase-vocoder.bundle.js' });
  if (!registered || registered.name !== 'phase-vocoder-processor') throw new Error('processor not registered');
  return registered.cls;
}

/* ---- seeded signals (SURVEY.md section 4); mirrored bit-for-bit by tests/signals.py ---- */
function lcgNoise(seed, n, amp) {
  const x = new Float32Array(n);
  let s = seed >>> 0;
  for (let i = 0; i < n; i++) {
    s = (Math.imul(s, 1664525) + 1013904223) >>> 0;
    x[i] = ((s >>> 8) - 8388608) / 8388608 * amp;
  }
  return x;
}
function tri(i, P) { return 4 * Math.abs((i % P) / P - 0.5) - 1; }
function makeSignal(kind, ch, n, fft) {
  if (kind === 'noise') return lcgNoise(1000 + ch, n, 0.5);
  if (kind === 'tonal') {
    const nz = lcgNoise(2000 + ch, n, 1 / 64);
    const x = new Float32Array(n);
    for (let i = 0; i < n; i++) x[i] = 0.25 * tri(i, 109) + 0.125 * tri(i, 31) + 0.0625 * tri(i, 7) + nz[i];
    return x;
  }
  if (kind === 'impulse') { const x = new Float32Array(n); x[1000 + ch] = 1.0; return x; }
  if (kind === 'sine32') {
    // 0.5*sin(2*pi*32*i/1024): only 32 distinct phases -> table lookup keeps it libm-independent across hosts
    const x = new Float32Array(n);
    for (let i = 0; i < n; i++) x[i] = 0.5 * Math.sin(2 * Math.PI * (i % 32) / 32);
    return x;
  }
  throw new Error('unknown signal ' + kind);
}
function pitchSchedule(spec, nhops) {
  const p = new Float32Array(nhops);
  if (spec.const !== undefined) p.fill(spec.const);
  else if (spec.sweep) { const [a, b, n] = spec.sweep; for (let i = 0; i < nhops; i++) p[i] = a + (b - a) * i / (n - 1); }
  else if (spec.list) { for (let i = 0; i < nhops; i++) p[i] = Number(spec.list[i]); }
  else throw new Error('bad pitch spec');
  return p;
}
const sha = (buf) => crypto.createHash('sha256').update(Buffer.from(buf.buffer, buf.byteOffset, buf.byteLength)).digest('hex');
function rms(x) { let s = 0; for (let i = 0; i < x.length; i++) s += x[i] * x[i]; return Math.sqrt(s / Math.max(1, x.length)); }

function runCase(c) {
  const N = c.fft, h = c.hop, T = c.nhops;
  const Cls = loadProcessorClass(N, h);
  const proc = new Cls({ numberOfInputs: 1, numberOfOutputs: 1 });
  const maxCh = Math.max(c.nch, ...(c.events || []).filter(e => e.type === 'channels').map(e => e.nch));
  const sig = []; for (let ch = 0; ch < maxCh; ch++) sig.push(makeSignal(c.signal, ch, T * h, N));
  const pitch = pitchSchedule(c.pitch, T);
  const out = []; for (let ch = 0; ch < maxCh; ch++) out.push(new Float32Array(T * h));
  const dumps = {};
  let nch = c.nch;
  let nout = -1;                       // 'out_channels' event: outputs[0].length differs from inputs[0].length from that hop on (-1: outputs mirror the inputs)
  for (let m = 0; m < T; m++) {
    let paused = false;
    for (const e of (c.events || [])) if (e.hop === m) { if (e.type === 'pause') paused = true; if (e.type === 'channels') nch = e.nch; if (e.type === 'out_channels') nout = e.nch; }
    const inputs = [[]], outputs = [[]];
    for (let ch = 0; ch < nch; ch++) {
      // host-owned blocks, valid only during the call (src/ola-processor.js:64)
      inputs[0].push(paused ? new Float32Array(0) : Float32Array.from(sig[ch].subarray(m * h, (m + 1) * h)));
      outputs[0].push(new Float32Array(h));
    }
    // the reference reallocates output buffers by outputs[0].length on its own (src/ola-processor.js:46-51) and writes input-count channels (:111-118)
    for (let ch = nch; ch < nout; ch++) outputs[0].push(new Float32Array(h).fill(123.0));
    let pf;
    if (c.arate) { pf = new Float32Array(h); pf.fill(0.7); pf[h - 1] = pitch[m]; } else pf = Float32Array.of(pitch[m]);
    const ret = proc.process(inputs, outputs, { pitchFactor: pf });
    if (ret !== true) throw new Error('process() did not return true');
    for (let ch = 0; ch < nch; ch++) out[ch].set(outputs[0][ch], m * h);
    if ((c.dump_hops || []).includes(m)) {
      dumps[m] = {
        X: Float64Array.from(proc.freqComplexBuffer),
        mag: Float32Array.from(proc.magnitudes),
        peaks: Int32Array.from(proc.peakIndexes.subarray(0, proc.nbPeaks)),
        Y: Float64Array.from(proc.freqComplexBufferShifted.slice(0, 2 * (N / 2 + 1))),
      };
    }
  }
  return { sig, pitch, out, dumps, maxCh };
}

/* numberOfInputs > 1 (src/ola-processor.js:10-11,24-33): every input has its own channel list, channel-count changes reallocate
 * only that input (src/ola-processor.js:38-52), timeCursor is shared (src/phase-vocoder.js:71).  Signals: channel c of input i uses
 * seed channel 10*i + c.  Output layout: input-major, each input padded to its largest channel count, store_hops hops per channel. */
function runMultiCase(c) {
  const N = c.fft, h = c.hop, T = c.nhops, nin = c.inputs.length;
  const Cls = loadProcessorClass(N, h);
  const proc = new Cls({ numberOfInputs: nin, numberOfOutputs: nin });
  const maxCh = c.inputs.map(inp => Math.max(inp.nch, ...(inp.events || []).filter(e => e.type === 'channels').map(e => e.nch)));
  const sig = c.inputs.map((inp, i) => { const a = []; for (let ch = 0; ch < maxCh[i]; ch++) a.push(makeSignal(c.signal, 10 * i + ch, T * h, N)); return a; });
  const out = c.inputs.map((inp, i) => { const a = []; for (let ch = 0; ch < maxCh[i]; ch++) a.push(new Float32Array(T * h)); return a; });
  const pitch = pitchSchedule(c.pitch, T);
  const nch = c.inputs.map(inp => inp.nch);
  for (let m = 0; m < T; m++) {
    const inputs = [], outputs = [];
    for (let i = 0; i < nin; i++) {
      for (const e of (c.inputs[i].events || [])) if (e.hop === m && e.type === 'channels') nch[i] = e.nch;
      const ins = [], outs = [];
      for (let ch = 0; ch < nch[i]; ch++) { ins.push(Float32Array.from(sig[i][ch].subarray(m * h, (m + 1) * h))); outs.push(new Float32Array(h)); }
      inputs.push(ins); outputs.push(outs);
    }
    if (proc.process(inputs, outputs, { pitchFactor: Float32Array.of(pitch[m]) }) !== true) throw new Error('process() did not return true');
    for (let i = 0; i < nin; i++) for (let ch = 0; ch < nch[i]; ch++) out[i][ch].set(outputs[i][ch], m * h);
  }
  const total = maxCh.reduce((a, b) => a + b, 0);
  const stored = new Float32Array(total * c.store_hops * h);
  let k = 0;
  for (let i = 0; i < nin; i++) for (let ch = 0; ch < maxCh[i]; ch++) stored.set(out[i][ch].subarray(0, c.store_hops * h), (k++) * c.store_hops * h);
  return { stored, pitch, maxCh, sig0: sig[0][0] };
}

function main() {
  const spec = JSON.parse(fs.readFileSync(path.join(HERE, 'cases.json'), 'utf8'));
  const manifest = { generator: 'tests/golden/gen_golden.js', node: process.version, reference_bundle_sha256: sha(fs.readFileSync(REF)), cases: [] };
  for (const c of spec.cases) {
    if (c.inputs) {
      const r = runMultiCase(c);
      fs.writeFileSync(path.join(HERE, c.name + '.out.f32'), Buffer.from(r.stored.buffer));
      manifest.cases.push(Object.assign({}, c, { out_file: c.name + '.out.f32', out_sha256: sha(r.stored), full_out_sha256: sha(r.stored), full_out_rms: rms(r.stored),
                                                  in_sha256_ch0: sha(r.sig0), pitch_sha256: sha(r.pitch), max_channels_per_input: r.maxCh, dumps: [] }));
      console.log(c.name.padEnd(40), 'out.sha', sha(r.stored).slice(0, 24), 'rms', rms(r.stored).toExponential(4));
      continue;
    }
    const r = runCase(c);
    const h = c.hop;
    // full-output fingerprint: all channels, all hops, channel-major (SURVEY.md section 4 convention)
    const full = new Float32Array(r.maxCh * c.nhops * h);
    for (let ch = 0; ch < r.maxCh; ch++) full.set(r.out[ch], ch * c.nhops * h);
    const stored = new Float32Array(c.store_ch * c.store_hops * h);
    for (let ch = 0; ch < c.store_ch; ch++) stored.set(r.out[ch].subarray(0, c.store_hops * h), ch * c.store_hops * h);
    fs.writeFileSync(path.join(HERE, c.name + '.out.f32'), Buffer.from(stored.buffer));
    const entry = Object.assign({}, c, {
      out_file: c.name + '.out.f32', out_sha256: sha(stored), full_out_sha256: sha(full), full_out_rms: rms(full),
      in_sha256_ch0: sha(r.sig[0]), pitch_sha256: sha(r.pitch), dumps: [],
    });
    for (const m of Object.keys(r.dumps)) {
      const d = r.dumps[m];
      const fn = c.name + '.dump' + m + '.bin';
      const parts = [d.X, d.mag, d.peaks, d.Y];
      fs.writeFileSync(path.join(HERE, fn), Buffer.concat(parts.map(p => Buffer.from(p.buffer, p.byteOffset, p.byteLength))));
      entry.dumps.push({ hop: Number(m), file: fn, layout: [['X', 'f64', d.X.length], ['mag', 'f32', d.mag.length], ['peaks', 'i32', d.peaks.length], ['Y', 'f64', d.Y.length]] });
    }
    manifest.cases.push(entry);
    console.log(c.name.padEnd(40), 'out.sha', entry.full_out_sha256.slice(0, 24), 'rms', entry.full_out_rms.toExponential(4), 'in.sha', entry.in_sha256_ch0.slice(0, 12));
  }
  fs.writeFileSync(path.join(HERE, 'manifest.json'), JSON.stringify(manifest, null, 1));
}
main();
