"use strict";
/*
 * sharded.js -- many independent streams on the GPUs of one node, driven from ONE Node.js process.
 *
 * Streams are independent processors (the reference builds one PhaseVocoderProcessor per stream; channels of a stream only share the
 * integer timeCursor: /root/reference/src/phase-vocoder.js:49-50,71), so the only multi-GPU structure there is to build is the partition:
 *
 *     stream s  ->  shard s mod G          (SURVEY 8e)      shard g = one native handle on device g mod deviceCount
 *
 * No data moves between GPUs.  Every shard owns ONE planar input and ONE planar output buffer in page-locked host memory (native.allocPinned ->
 * pv_host_alloc): a host writes the samples of stream s, channel c straight into inputView(s, c), calls processInPlace(nhops) and reads
 * outputView(s, c) -- no re-packing in JavaScript, and the native batch call pipelines pinned buffers (H2D of piece k+1, kernel of piece k and
 * D2H of piece k-1 at the same time).  processInPlace starts ALL shards through the asynchronous entry of the addon (napi_async_work ->
 * pv_process_batch on a libuv worker thread, one per shard) and only then awaits them: every GPU has its batch in flight before the first
 * wait.  processBatch(inputs, outputs, ...) keeps the round-3 form for hosts whose samples live in their own arrays (one copy in, one copy out).
 * With fewer devices than requested shards the shards share devices (each handle has its own HIP stream) and the result says so
 * (`replicasMeasured`), exactly as bench.py does.
 *
 * libuv starts 4 worker threads by default: for more than 4 shards set UV_THREADPOOL_SIZE >= shards BEFORE Node starts any thread-pool
 * work (the constructor sets it when it still can and reports `threadPoolSize`).
 */
const path = require("path");
const native = require(path.join(__dirname, "phaze_napi.node"));

class ShardedPhaseVocoder {
    /**
     * @param {object} o  fftSize, hopSize (reference defaults 2048 / 128), channelsPerStream, streams (total), maxHops (per batch),
     *                    gpus (requested shards; default = devices present), flags
     */
    constructor(o) {
        o = o || {};
        this.fftSize = o.fftSize !== undefined ? o.fftSize : 2048;          // phase-vocoder.js:6
        this.hopSize = o.hopSize !== undefined ? o.hopSize : 128;           // ola-processor.js:3
        this.channelsPerStream = o.channelsPerStream || 1;
        this.streams = o.streams || 1;
        this.maxHops = o.maxHops || 1;
        this.devicesPresent = native.deviceCount();
        if (this.devicesPresent < 1) throw new Error("no HIP device available (this library has no CPU path)");
        this.requestedGpus = o.gpus || this.devicesPresent;
        this.shards = Math.min(this.requestedGpus, this.streams);
        if (!process.env.UV_THREADPOOL_SIZE || (process.env.UV_THREADPOOL_SIZE | 0) < this.shards) process.env.UV_THREADPOOL_SIZE = String(Math.max(4, this.shards));
        this.threadPoolSize = process.env.UV_THREADPOOL_SIZE | 0;
        this._streamsOf = [];
        this._handles = [];
        this._in = [];
        this._out = [];
        this._pitch = [];
        for (let g = 0; g < this.shards; g++) {
            const mine = [];
            for (let s = g; s < this.streams; s += this.shards) mine.push(s);   // stream s -> shard s mod G
            this._streamsOf.push(mine);
            const nch = mine.length * this.channelsPerStream;
            this._handles.push(native.create({ fftSize: this.fftSize, hopSize: this.hopSize, maxChannels: nch, maxHops: this.maxHops,
                                               deviceId: g % this.devicesPresent, flags: o.flags | 0 }));
            // page-locked, planar: channel slot q of the shard at [q * stride, (q + 1) * stride), stride = maxHops * hopSize floats
            this._in.push(native.allocPinned(nch * this.maxHops * this.hopSize));
            this._out.push(native.allocPinned(nch * this.maxHops * this.hopSize));
            this._pitch.push(new Float32Array(mine.length * this.maxHops).fill(1.0));
        }
        this.stride = this.maxHops * this.hopSize;
    }

    /** The samples of stream s, channel c live HERE (maxHops * hopSize floats of the shard's pinned input / output buffer): write / read in place. */
    inputView(s, c) { const w = this.shardOf(s), q = w.slot * this.channelsPerStream + c; return this._in[w.shard].subarray(q * this.stride, (q + 1) * this.stride); }
    outputView(s, c) { const w = this.shardOf(s), q = w.slot * this.channelsPerStream + c; return this._out[w.shard].subarray(q * this.stride, (q + 1) * this.stride); }
    /** k-rate pitchFactor of stream s per hop (phase-vocoder.js:47), maxHops floats, defaults to 1.0 */
    pitchView(s) { const w = this.shardOf(s); return this._pitch[w.shard].subarray(w.slot * this.maxHops, (w.slot + 1) * this.maxHops); }

    /** nhops consecutive process() calls for every stream, on the views above.  Resolves when every output view is complete. */
    async processInPlace(nhops) {
        if (nhops > this.maxHops) throw new Error("processInPlace: nhops exceeds maxHops");
        const jobs = [];
        for (let g = 0; g < this.shards; g++) {                              // launch every shard ...
            const nch = this._streamsOf[g].length * this.channelsPerStream;
            jobs.push(native.processBatchAsync(this._handles[g], this._in[g], this._out[g], nch, nhops, this._pitch[g], this.maxHops, this.channelsPerStream, this.stride));
        }
        await Promise.all(jobs);                                             // ... before the first wait
        return true;
    }

    /** shard (= handle) that owns stream s, and its slot there */
    shardOf(s) { return { shard: s % this.shards, slot: Math.floor(s / this.shards) }; }

    get replicasMeasured() { return Math.min(this.shards, this.devicesPresent); }

    /**
     * nhops consecutive process() calls for every stream.  inputs[s][c] / outputs[s][c]: Float32Array(nhops * hopSize) of stream s, channel c;
     * pitch[s]: Float32Array(nhops), the k-rate pitchFactor of stream s per hop (phase-vocoder.js:47).  Resolves when every output is written.
     */
    async processBatch(inputs, outputs, pitch, nhops) {
        if (nhops > this.maxHops) throw new Error("processBatch: nhops exceeds maxHops");
        const n = nhops * this.hopSize, cps = this.channelsPerStream;
        for (let s = 0; s < this.streams; s++) {
            for (let c = 0; c < cps; c++) this.inputView(s, c).set(inputs[s][c].subarray(0, n));
            this.pitchView(s).set(pitch[s].subarray(0, nhops));
        }
        await this.processInPlace(nhops);
        for (let s = 0; s < this.streams; s++)
            for (let c = 0; c < cps; c++) outputs[s][c].set(this.outputView(s, c).subarray(0, n));
        return true;
    }

    /** How many shards' last batches were in flight TOGETHER (largest set of pairwise overlapping worker-thread windows): equals `shards` when the
     *  libuv pool has a thread for each (UV_THREADPOOL_SIZE >= shards took effect), at most the pool size otherwise. */
    maxConcurrentShards() {
        const w = this._handles.map((h) => native.batchWindow(h));
        let best = 0;
        for (const [b] of w) { let n = 0; for (const [b2, e2] of w) if (b2 <= b && b < e2) n++; if (n > best) best = n; }   // windows open at the instant a window opens
        return best;
    }

    /** state of channel c of stream s: {hist, acc, timeCursor} (checkpoint / resume / moving a stream to another shard) */
    exportState(s, c) { const w = this.shardOf(s); return native.exportState(this._handles[w.shard], w.slot * this.channelsPerStream + c); }
    importState(s, c, st) { const w = this.shardOf(s); native.importState(this._handles[w.shard], w.slot * this.channelsPerStream + c, st.hist || null, st.acc || null); }

    info() { return this._handles.map((h) => native.info(h)); }

    // (the page-locked shard buffers go with their last reference: dropped here, so that a host that creates and closes sharded processors does not keep them pinned)
    close() { for (const h of this._handles) native.destroy(h); this._handles = []; this._in = []; this._out = []; this._pitch = []; }
}

module.exports = { ShardedPhaseVocoder };
