"use strict";
/*
 * sharded.js -- many independent streams on the GPUs of one node, driven from ONE Node.js process.
 *
 * Streams are independent processors (the reference builds one PhaseVocoderProcessor per stream; channels of a stream only share the
 * integer timeCursor: /root/reference/src/phase-vocoder.js:49-50,71), so the only multi-GPU structure there is to build is the partition:
 *
 *     stream s  ->  shard s mod G          (SURVEY 8e)      shard g = one native handle on device g mod deviceCount
 *
 * No data moves between GPUs.  processBatch() packs every shard's streams into one planar buffer, starts ALL shards through the
 * asynchronous entry of the addon (napi_async_work -> pv_process_batch on a libuv worker thread, one per shard) and only then awaits them:
 * every GPU has its batch in flight before the first wait.  With fewer devices than requested shards the shards share devices (each handle
 * has its own HIP stream) and the result says so (`replicasMeasured`), exactly as bench.py does.
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
            this._in.push(new Float32Array(nch * this.maxHops * this.hopSize));
            this._out.push(new Float32Array(nch * this.maxHops * this.hopSize));
            this._pitch.push(new Float32Array(mine.length * this.maxHops));
        }
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
        const jobs = [];
        for (let g = 0; g < this.shards; g++) {                              // pack + launch every shard ...
            const mine = this._streamsOf[g], nch = mine.length * cps;
            const bin = this._in[g].subarray(0, nch * n), bout = this._out[g].subarray(0, nch * n), bp = this._pitch[g].subarray(0, mine.length * nhops);
            for (let k = 0; k < mine.length; k++) {
                for (let c = 0; c < cps; c++) bin.set(inputs[mine[k]][c].subarray(0, n), (k * cps + c) * n);
                bp.set(pitch[mine[k]].subarray(0, nhops), k * nhops);
            }
            jobs.push(native.processBatchAsync(this._handles[g], bin, bout, nch, nhops, bp, nhops, cps));
        }
        await Promise.all(jobs);                                             // ... before the first wait
        for (let g = 0; g < this.shards; g++) {
            const mine = this._streamsOf[g];
            for (let k = 0; k < mine.length; k++)
                for (let c = 0; c < cps; c++) outputs[mine[k]][c].set(this._out[g].subarray((k * cps + c) * n, (k * cps + c + 1) * n));
        }
        return true;
    }

    /** state of channel c of stream s: {hist, acc, timeCursor} (checkpoint / resume / moving a stream to another shard) */
    exportState(s, c) { const w = this.shardOf(s); return native.exportState(this._handles[w.shard], w.slot * this.channelsPerStream + c); }
    importState(s, c, st) { const w = this.shardOf(s); native.importState(this._handles[w.shard], w.slot * this.channelsPerStream + c, st.hist || null, st.acc || null); }

    info() { return this._handles.map((h) => native.info(h)); }

    close() { for (const h of this._handles) native.destroy(h); this._handles = []; }
}

module.exports = { ShardedPhaseVocoder };
