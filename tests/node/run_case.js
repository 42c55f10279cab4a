#!/usr/bin/env node
/*
 * run_case.js -- drives the Node.js host (phaze_amd/node/phase-vocoder.js) exactly like the golden generator
 * drives the reference: hop-sized planar blocks, 1-element (or a-rate) pitchFactor arrays, pause / channel
 * change events.  Inputs come from files written by pytest; outputs go back to a file.
 * usage: node run_case.js <spec.json>
 */
"use strict";
const fs = require("fs");
const path = require("path");
const { PhaseVocoderProcessor, getProcessor } = require(path.join(__dirname, "..", "..", "phaze_amd", "node", "phase-vocoder.js"));

const spec = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
const N = spec.fft, h = spec.hop, T = spec.nhops, maxCh = spec.max_ch;
const rd = (f) => { const b = fs.readFileSync(f); return new Float32Array(b.buffer, b.byteOffset, b.byteLength / 4); };
const x = rd(spec.in_file), pitch = rd(spec.pitch_file);
const Cls = getProcessor("phase-vocoder-processor");
if (Cls !== PhaseVocoderProcessor) throw new Error("registration broken");
const opts = { numberOfInputs: 1, numberOfOutputs: 1 };
if (!(N === 2048 && h === 128 && spec.use_defaults)) opts.processorOptions = { fftSize: N, hopSize: h };
if (spec.flags) opts.processorOptions = Object.assign(opts.processorOptions || {}, { flags: spec.flags });   // PV_FLAG_* (32: resident streaming kernel)
const proc = new Cls(opts);
const out = new Float32Array(maxCh * T * h);
let nch = spec.nch, nout = -1;
const t0 = process.hrtime.bigint();
for (let m = 0; m < T; m++) {
  let paused = false;
  for (const e of (spec.events || [])) if (e.hop === m) { if (e.type === "pause") paused = true; if (e.type === "channels") nch = e.nch; if (e.type === "out_channels") nout = e.nch; }
  const inputs = [[]], outputs = [[]];
  for (let c = 0; c < nch; c++) {
    inputs[0].push(paused ? new Float32Array(0) : x.subarray(c * T * h + m * h, c * T * h + (m + 1) * h));
    outputs[0].push(new Float32Array(h));
  }
  let pf;
  if (spec.arate) { pf = new Float32Array(h); pf.fill(0.7); pf[h - 1] = pitch[m]; } else pf = Float32Array.of(pitch[m]);
  for (let c = nch; c < nout; c++) outputs[0].push(new Float32Array(h));        // outputs that do not mirror the inputs (ola-processor.js:46-51)
  if (proc.process(inputs, outputs, { pitchFactor: pf }) !== true) throw new Error("process() must return true");
  for (let c = 0; c < nch; c++) out.set(outputs[0][c], c * T * h + m * h);
}
const dt = Number(process.hrtime.bigint() - t0) / 1e9;
if (proc.timeCursor !== T * h) throw new Error("timeCursor " + proc.timeCursor + " != " + T * h);
fs.writeFileSync(spec.out_file, Buffer.from(out.buffer));
console.log(JSON.stringify({ ok: true, calls: T, seconds: dt, us_per_call: dt / T * 1e6, info: proc.info() }));
proc.close();
