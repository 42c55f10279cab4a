#!/usr/bin/env node
"use strict";
/*
 * pitch-shift-cli.js -- offline WAV-in / WAV-out pitch shifter on top of the Node host (phase-vocoder.js).
 *
 * Counterpart of the reference demo's control path (/root/reference/src/main.js:39-50,75-96): there a PlayerEngine plays
 * the file at `speed` (which also scales pitch by `speed`) and the worklet corrects with pitchFactor = pitch / speed
 * (main.js:82,93).  Here the "player" is a linear-interpolation resampler and the worklet is the GPU processor:
 *
 *     node pitch-shift-cli.js in.wav out.wav [--pitch 1.5] [--speed 1.0] [--fft 2048] [--hop 128] [--batch]
 *
 * --batch uses the throughput entry point (all hops in one launch); default drives process() one render quantum at a
 * time exactly like an audio graph would.  PCM16 / PCM24 / float32 RIFF WAV in, float32 WAV out.
 */
const fs = require("fs");
const path = require("path");
const { PhaseVocoderProcessor } = require(path.join(__dirname, "phase-vocoder.js"));

function parseWav(buf) {
  if (buf.toString("ascii", 0, 4) !== "RIFF" || buf.toString("ascii", 8, 12) !== "WAVE") throw new Error("not a RIFF/WAVE file");
  let pos = 12, fmt = null, data = null;
  while (pos + 8 <= buf.length) {
    const id = buf.toString("ascii", pos, pos + 4), size = buf.readUInt32LE(pos + 4);
    if (id === "fmt ") fmt = { tag: buf.readUInt16LE(pos + 8), ch: buf.readUInt16LE(pos + 10), rate: buf.readUInt32LE(pos + 12), bits: buf.readUInt16LE(pos + 22) };
    if (id === "data") data = buf.subarray(pos + 8, pos + 8 + size);
    pos += 8 + size + (size & 1);
  }
  if (!fmt || !data) throw new Error("missing fmt/data chunk");
  const bps = fmt.bits / 8, n = Math.floor(data.length / (bps * fmt.ch));
  const chans = Array.from({ length: fmt.ch }, () => new Float32Array(n));
  for (let i = 0; i < n; i++) for (let c = 0; c < fmt.ch; c++) {
    const o = (i * fmt.ch + c) * bps;
    let v;
    if (fmt.tag === 3 && fmt.bits === 32) v = data.readFloatLE(o);
    else if (fmt.bits === 16) v = data.readInt16LE(o) / 32768;
    else if (fmt.bits === 24) v = data.readIntLE(o, 3) / 8388608;
    else if (fmt.bits === 32) v = data.readInt32LE(o) / 2147483648;
    else throw new Error("unsupported sample format");
    chans[c][i] = v;
  }
  return { rate: fmt.rate, chans };
}

function writeWavF32(file, rate, chans) {
  const n = chans[0].length, ch = chans.length, buf = Buffer.alloc(44 + n * ch * 4);
  buf.write("RIFF", 0); buf.writeUInt32LE(36 + n * ch * 4, 4); buf.write("WAVEfmt ", 8); buf.writeUInt32LE(16, 16);
  buf.writeUInt16LE(3, 20); buf.writeUInt16LE(ch, 22); buf.writeUInt32LE(rate, 24); buf.writeUInt32LE(rate * ch * 4, 28);
  buf.writeUInt16LE(ch * 4, 32); buf.writeUInt16LE(32, 34); buf.write("data", 36); buf.writeUInt32LE(n * ch * 4, 40);
  for (let i = 0; i < n; i++) for (let c = 0; c < ch; c++) buf.writeFloatLE(chans[c][i], 44 + (i * ch + c) * 4);
  fs.writeFileSync(file, buf);
}

function resample(x, speed) {            // the "PlayerEngine": play x at `speed`
  if (speed === 1) return x;
  const n = Math.floor(x.length / speed), y = new Float32Array(n);
  for (let i = 0; i < n; i++) { const t = i * speed, k = Math.floor(t), f = t - k; y[i] = x[k] * (1 - f) + (k + 1 < x.length ? x[k + 1] : 0) * f; }
  return y;
}

function main() {
  const a = process.argv.slice(2), opt = { pitch: 1, speed: 1, fft: 2048, hop: 128, batch: false }, files = [];
  for (let i = 0; i < a.length; i++) {
    if (a[i] === "--batch") opt.batch = true;
    else if (a[i].startsWith("--")) opt[a[i].slice(2)] = Number(a[++i]);
    else files.push(a[i]);
  }
  if (files.length !== 2) { console.error("usage: pitch-shift-cli.js in.wav out.wav [--pitch P] [--speed S] [--fft N] [--hop H] [--batch]"); process.exit(2); }
  const wav = parseWav(fs.readFileSync(files[0]));
  const pitchFactor = opt.pitch / opt.speed;                             // main.js:82,93
  const ins = wav.chans.map((c) => resample(c, opt.speed));
  const nch = ins.length, h = opt.hop, latency = opt.fft - h;            // output delay N - hop (K2)
  const nhops = Math.ceil((ins[0].length + latency) / h);
  const proc = new PhaseVocoderProcessor({ numberOfInputs: 1, numberOfOutputs: 1, processorOptions: { fftSize: opt.fft, hopSize: h, maxHops: opt.batch ? nhops : 1 } });
  const padded = ins.map((c) => { const p = new Float32Array(nhops * h); p.set(c); return p; });
  const outs = Array.from({ length: nch }, () => new Float32Array(nhops * h));
  const t0 = process.hrtime.bigint();
  if (opt.batch) {
    const flatIn = new Float32Array(nch * nhops * h), flatOut = new Float32Array(nch * nhops * h);
    padded.forEach((c, i) => flatIn.set(c, i * nhops * h));
    proc.processBatch(flatIn, flatOut, nch, nhops, new Float32Array(nhops).fill(pitchFactor));
    outs.forEach((o, i) => o.set(flatOut.subarray(i * nhops * h, (i + 1) * nhops * h)));
  } else {
    const pf = Float32Array.of(pitchFactor);
    for (let m = 0; m < nhops; m++) {
      const inputs = [padded.map((c) => c.subarray(m * h, (m + 1) * h))], outputs = [outs.map((o) => o.subarray(m * h, (m + 1) * h))];
      proc.process(inputs, outputs, { pitchFactor: pf });
    }
  }
  const secs = Number(process.hrtime.bigint() - t0) / 1e9;
  // drop the N - hop samples of algorithmic delay and undo the 3R/8R Hann^2 overlap gain (0.375 for R >= 4: K1)
  const gain = 1 / 0.375;
  const trimmed = outs.map((o) => o.subarray(latency, latency + ins[0].length).map((v) => v * gain));
  writeWavF32(files[1], wav.rate, trimmed);
  console.log(JSON.stringify({ ok: true, channels: nch, rate: wav.rate, frames: nhops * nch, pitchFactor, seconds: secs, mode: opt.batch ? "batch" : "streaming", device: proc.info().deviceName }));
  proc.close();
}
main();
