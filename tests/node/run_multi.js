#!/usr/bin/env node
/*
 * run_multi.js -- drives the Node.js host with numberOfInputs > 1 exactly like the golden generator drives the reference
 * (tests/golden/gen_golden.js runMultiCase): per-input channel lists, channel-count changes per input, one shared pitchFactor.
 * usage: node run_multi.js <spec.json>     (spec: fft, hop, nhops, inputs[{nch, events, max_ch, in_file}], pitch_file, out_file)
 */
"use strict";
const fs = require("fs");
const path = require("path");
const { getProcessor } = require(path.join(__dirname, "..", "..", "phaze_amd", "node", "phase-vocoder.js"));
const spec = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
const N = spec.fft, h = spec.hop, T = spec.nhops, nin = spec.inputs.length;
const rd = (f) => { const b = fs.readFileSync(f); return new Float32Array(b.buffer, b.byteOffset, b.byteLength / 4); };
const pitch = rd(spec.pitch_file);
const x = spec.inputs.map((inp) => rd(inp.in_file));                       // [max_ch][T*h] per input
const Cls = getProcessor("phase-vocoder-processor");
const proc = new Cls({ numberOfInputs: nin, numberOfOutputs: nin, processorOptions: { fftSize: N, hopSize: h, flags: spec.flags | 0 } });   // flags 32: one resident kernel per input
const out = spec.inputs.map((inp) => new Float32Array(inp.max_ch * T * h));
const nch = spec.inputs.map((inp) => inp.nch);
for (let m = 0; m < T; m++) {
  const inputs = [], outputs = [];
  for (let i = 0; i < nin; i++) {
    for (const e of (spec.inputs[i].events || [])) if (e.hop === m && e.type === "channels") nch[i] = e.nch;
    const ins = [], outs = [];
    for (let c = 0; c < nch[i]; c++) { ins.push(x[i].subarray(c * T * h + m * h, c * T * h + (m + 1) * h)); outs.push(new Float32Array(h)); }
    inputs.push(ins); outputs.push(outs);
  }
  if (proc.process(inputs, outputs, { pitchFactor: Float32Array.of(pitch[m]) }) !== true) throw new Error("process() must return true");
  for (let i = 0; i < nin; i++) for (let c = 0; c < nch[i]; c++) out[i].set(outputs[i][c], c * T * h + m * h);
}
if (proc.timeCursor !== T * h) throw new Error("timeCursor " + proc.timeCursor + " != " + T * h);
fs.writeFileSync(spec.out_file, Buffer.concat(out.map((o) => Buffer.from(o.buffer))));
console.log(JSON.stringify({ ok: true, calls: T }));
proc.close();
