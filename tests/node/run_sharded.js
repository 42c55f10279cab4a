#!/usr/bin/env node
/*
 * run_sharded.js -- drives phaze_amd/node/sharded.js for the GPU tests: S streams x C channels split over `shards` native handles (on one GPU they
 * share device 0: two handles in flight concurrently), against (a) ONE handle fed the same streams and (b) the caller's reference file.
 * Also exercises the round-3 addon entries: the busy guard of processBatchAsync and exportState / importState through Node.
 * usage: node run_sharded.js <spec.json>   (spec: fft, hop, nhops, streams, cps, shards, in_file [S*C][T*h], pitch_file [S][T], out_file)
 */
"use strict";
const fs = require("fs");
const path = require("path");
const { ShardedPhaseVocoder } = require(path.join(__dirname, "..", "..", "phaze_amd", "node", "sharded.js"));
const { native } = require(path.join(__dirname, "..", "..", "phaze_amd", "node", "phase-vocoder.js"));
const spec = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
const { fft, hop, nhops: T, streams: S, cps, shards } = spec;
const rd = (f) => { const b = fs.readFileSync(f); return new Float32Array(b.buffer, b.byteOffset, b.byteLength / 4); };
const x = rd(spec.in_file), p = rd(spec.pitch_file), n = T * hop;

(async () => {
  const res = {};
  // (1) sharded: every shard in flight before the first wait
  const pv = new ShardedPhaseVocoder({ fftSize: fft, hopSize: hop, channelsPerStream: cps, streams: S, maxHops: T, gpus: shards });
  const inputs = [], outputs = [], pitch = [];
  for (let s = 0; s < S; s++) {
    inputs.push([]); outputs.push([]); pitch.push(p.subarray(s * T, (s + 1) * T));
    for (let c = 0; c < cps; c++) { inputs[s].push(x.subarray((s * cps + c) * n, (s * cps + c + 1) * n)); outputs[s].push(new Float32Array(n)); }
  }
  // two half-batches: state (history, accumulator, timeCursor) carries across calls in every shard
  const T1 = Math.floor(T / 2), cut = (a, lo, hi) => a.map((st) => st.map((ch) => ch.subarray(lo * hop, hi * hop)));
  await pv.processBatch(cut(inputs, 0, T1), cut(outputs, 0, T1), pitch.map((q) => q.subarray(0, T1)), T1);
  await pv.processBatch(cut(inputs, T1, T), cut(outputs, T1, T), pitch.map((q) => q.subarray(T1, T)), T - T1);
  res.shards = pv.shards; res.requested = pv.requestedGpus; res.replicas = pv.replicasMeasured; res.devices = pv.devicesPresent;
  const out = new Float32Array(S * cps * n);
  for (let s = 0; s < S; s++) for (let c = 0; c < cps; c++) out.set(outputs[s][c], (s * cps + c) * n);
  fs.writeFileSync(spec.out_file, Buffer.from(out.buffer));
  // (2) ONE handle, synchronous batch, same streams: must agree bit for bit (streams are independent, K5)
  const h1 = native.create({ fftSize: fft, hopSize: hop, maxChannels: S * cps, maxHops: T });
  const one = new Float32Array(S * cps * n);
  native.processBatch(h1, x, one, S * cps, T, p, T, cps);
  res.equal_to_one_handle = Buffer.compare(Buffer.from(out.buffer), Buffer.from(one.buffer)) === 0;
  // (3) busy guard: a second call on a handle whose asynchronous batch is in flight is refused, the batch still completes
  const again = new Float32Array(S * cps * n);
  native.reset(h1);
  const pr = native.processBatchAsync(h1, x, again, S * cps, T, p, T, cps);
  try { native.timeCursor(h1); res.busy_guard = "no throw"; } catch (e) { res.busy_guard = e.code; }
  await pr;
  res.async_equals_sync = Buffer.compare(Buffer.from(again.buffer), Buffer.from(one.buffer)) === 0;
  // (4) exportState / importState through Node: hand stream 0 (all its channels) from the sharded object to a fresh handle mid-stream
  const st = []; for (let c = 0; c < cps; c++) st.push(pv.exportState(0, c));
  res.state_cursor = st[0].timeCursor; res.state_len = st[0].hist.length;
  const h2 = native.create({ fftSize: fft, hopSize: hop, maxChannels: cps, maxHops: T });
  for (let c = 0; c < cps; c++) native.importState(h2, c, st[c].hist, st[c].acc, st[c].timeCursor);
  const ext = new Float32Array(cps * n), cont = new Float32Array(cps * n);           // continue stream 0 with its own first T hops again, on both
  for (let c = 0; c < cps; c++) ext.set(inputs[0][c], c * n);
  native.processBatch(h2, ext, cont, cps, T, pitch[0], 0, 1);
  const o2 = []; for (let s = 0; s < S; s++) { o2.push([]); for (let c = 0; c < cps; c++) o2[s].push(new Float32Array(n)); }
  await pv.processBatch(inputs, o2, pitch, T);
  let same = true; for (let c = 0; c < cps && same; c++) same = Buffer.compare(Buffer.from(o2[0][c].buffer), Buffer.from(cont.buffer, c * n * 4, n * 4)) === 0;
  res.migrated_stream_continues_bit_exact = same;
  // (5) round 4: the in-place form -- streams written straight into the shards' pinned buffers (inputView / pitchView), processInPlace, results read
  // from outputView -- gives the bits of the copying form; and every shard's batch was in flight at the same time (libuv pool >= shards)
  const pv2 = new ShardedPhaseVocoder({ fftSize: fft, hopSize: hop, channelsPerStream: cps, streams: S, maxHops: T + 3, gpus: shards });   // maxHops > nhops: strided rows
  for (let s = 0; s < S; s++) { for (let c = 0; c < cps; c++) pv2.inputView(s, c).set(inputs[s][c]); pv2.pitchView(s).set(pitch[s]); }
  await pv2.processInPlace(T);
  let inplace = true;
  for (let s = 0; s < S && inplace; s++) for (let c = 0; c < cps && inplace; c++)
    inplace = Buffer.compare(Buffer.from(pv2.outputView(s, c).buffer, pv2.outputView(s, c).byteOffset, n * 4), Buffer.from(one.buffer, (s * cps + c) * n * 4, n * 4)) === 0;
  res.in_place_equals_one_handle = inplace;
  res.shards_in_flight_together = pv2.maxConcurrentShards();
  res.thread_pool = pv2.threadPoolSize;
  pv2.close();
  native.destroy(h1); native.destroy(h2); pv.close();
  console.log(JSON.stringify(res));
})().catch((e) => { console.error(e); process.exit(1); });
