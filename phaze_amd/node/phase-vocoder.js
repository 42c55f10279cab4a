"use strict";
/*
 * phase-vocoder.js -- Node.js host for the MI355X-native pitch shifter.
 *
 * Drop-in for the reference worklet module (/root/reference/src/phase-vocoder.js + src/ola-processor.js):
 * same registration name, same `parameterDescriptors`, same constructor shape, same
 * `process(inputs, outputs, parameters) -> true`.  The per-frame work (Hann -> real FFT -> peak picking ->
 * region shift -> inverse FFT -> Hann -> overlap-add) runs as HIP kernels behind the N-API addon; this file
 * only keeps the bookkeeping the reference does in JS around that call:
 *   - one native handle per input (channels of one input share a launch),
 *   - channel (re)allocation as the reference does it, inputs and outputs SEPARATELY (ola-processor.js:38-52): a changed inputs[i].length restarts the
 *     input history of that input from zeros (:54-71), a changed outputs[i].length its pending overlap-add sums (:73-88); only the input-count
 *     channels are processed and written (phase-vocoder.js:49-51, ola-processor.js:111-118),
 *   - the paused branch (ola-processor.js:93-100),
 *   - k-rate pitchFactor = last element of the parameter array (phase-vocoder.js:47).
 * There is no JS fallback for the DSP: without the addon / a GPU, construction throws.
 */
const path = require("path");
const native = require(path.join(__dirname, "phaze_napi.node"));

const HOST_CHANNEL_BOOKKEEPING = 128, FP64_FORWARD = 256, STATE_HISTORY = 1, STATE_ACCUMULATOR = 2;   // PV_FLAG_HOST_CHANNEL_BOOKKEEPING, PV_STATE_* of include/phaze_amd.h
const BUFFERED_BLOCK_SIZE = 2048;   // reference default (phase-vocoder.js:6)
const WEBAUDIO_BLOCK_SIZE = 128;    // reference default (ola-processor.js:3)

// Host globals of an AudioWorkletGlobalScope; Node has neither, so provide the minimal equivalents.
const Base = (typeof AudioWorkletProcessor !== "undefined") ? AudioWorkletProcessor : class AudioWorkletProcessor {
    constructor(options) { this.port = null; }
};
const registry = new Map();
const register = (typeof registerProcessor !== "undefined") ? registerProcessor : function registerProcessor(name, cls) {
    if (registry.has(name)) throw new Error(`processor '${name}' is already registered`);
    registry.set(name, cls);
};

class PhaseVocoderProcessor extends Base {
    static get parameterDescriptors() {                       // phase-vocoder.js:17-22
        return [{ name: "pitchFactor", defaultValue: 1.0 }];
    }

    constructor(options) {
        super(options);
        options = options || {};
        const po = options.processorOptions || {};
        this.nbInputs = options.numberOfInputs;                // required, as in ola-processor.js:10-11
        this.nbOutputs = options.numberOfOutputs;
        this.blockSize = po.fftSize !== undefined ? po.fftSize : BUFFERED_BLOCK_SIZE;
        this.hopSize = po.hopSize !== undefined ? po.hopSize : WEBAUDIO_BLOCK_SIZE;
        this.fftSize = this.blockSize;
        this.nbOverlaps = this.blockSize / this.hopSize;       // ola-processor.js:17
        this.deviceId = po.deviceId | 0;
        this._flags = po.flags | 0;                                     // PV_FLAG_* of include/phaze_amd.h (e.g. 32 = resident streaming kernel); 0 in normal use
        this._maxHops = Math.max(1, po.maxHops | 0);                   // staging size of the throughput entry point (processBatch)
        this._handles = [];
        this._channels = [];                                            // inputs[i].length the input side was "allocated" for
        this._outChannels = [];                                         // outputs[i].length the output side was "allocated" for
        this._capacity = [];
        // what it takes to reproduce an output channel that lost its input while outputs[i].length stayed (reallocateChannelsIfNeeded, _staleFrame):
        this._rings = [];                                               // per input, per channel slot: the last blockSize input samples as a ring (inputBuffers, ola-processor.js:59,121-127)
        this._ringPos = [];                                             // per input: where the next block goes
        this._stale = [];                                               // per input: Map channel -> { frame: frame / nbOverlaps as the last real quantum added it, q: quanta it has been re-added }
        this._lastPitch = undefined;
        this._scratch = null;
        this._flags |= HOST_CHANNEL_BOOKKEEPING;                        // the resets below replace the C ABI's own (mirrored) channel-count rule
        for (let i = 0; i < (this.nbInputs | 0); i++) {
            // "default to 1 channel per input until we know more" (ola-processor.js:24-27); capacity 2 avoids a
            // re-create for the common mono->stereo switch.  Throws Error('FFT size must be a power of two and
            // bigger than 1') for bad sizes, as `new FFT(n)` does (bundle:6-7).
            this._capacity.push(2);
            this._handles.push(native.create({ fftSize: this.blockSize, hopSize: this.hopSize, maxChannels: 2, maxHops: this._maxHops, deviceId: this.deviceId, flags: this._flags }));
            this._channels.push(1);
            this._outChannels.push(1);
            this._rings.push([new Float32Array(this.blockSize), new Float32Array(this.blockSize)]);
            this._ringPos.push(0);
            this._stale.push(new Map());
        }
    }

    /** The last blockSize input samples of channel c of input i, oldest first (the reference's inputBuffers after shiftInputBuffers). */
    _window(i, c) {
        const N = this.blockSize, ring = this._rings[i][c], pos = this._ringPos[i], w = new Float32Array(N);
        w.set(ring.subarray(pos), 0);
        w.set(ring.subarray(0, pos), N - pos);
        return w;
    }

    /** The windowed frame / nbOverlaps the last quantum added for channel c of input i, recomputed on a one-channel scratch handle from that window with a zero
     *  accumulator: the hop that comes out and the accumulator left behind ARE the frame (0 + x is exact). */
    _staleFrame(i, c) {
        const N = this.blockSize, h = this.hopSize, t = native.timeCursor(this._handles[i]);
        if (this._lastPitch === undefined || t < h) return new Float32Array(N);      // no frame yet: outputBuffersToRetrieve still holds its zeros
        if (!this._scratch) this._scratch = native.create({ fftSize: N, hopSize: h, maxChannels: 1, maxHops: 1, deviceId: this.deviceId, flags: this._flags & FP64_FORWARD });
        const w = this._window(i, c), out = [new Float32Array(h)];
        native.importState(this._scratch, 0, w.slice(0, N - h), new Float32Array(N - h), t - h);
        native.process(this._scratch, [w.slice(N - h)], out, this._lastPitch);
        const frame = new Float32Array(N);
        frame.set(out[0], 0);
        if (N > h) frame.set(native.exportState(this._scratch, 0).acc, h);
        return frame;
    }

    /** outputBuffers of a channel whose stale frame has been re-added for q quanta: f32 adds in the reference's order (ola-processor.js:130-157). */
    _staleAccumulator(i, c) {
        const N = this.blockSize, h = this.hopSize, rec = this._stale[i].get(c), a = new Float32Array(N);
        if (N > h) a.set(native.exportState(this._handles[i], c).acc, 0);
        for (let n = Math.min(rec.q, N / h + 1); n > 0; n--) {                         // (after N / hop quanta the sums no longer change)
            for (let k = 0; k < N; k++) a[k] += rec.frame[k];
            a.copyWithin(0, h);
            a.fill(0, N - h);
        }
        return a.slice(0, N - h);
    }

    get timeCursor() { return this._handles.length ? native.timeCursor(this._handles[0]) : 0; }   // phase-vocoder.js:31

    /** Handles dynamic reallocation of input/output channels buffer (ola-processor.js:38-52): inputs and outputs are two events. */
    reallocateChannelsIfNeeded(inputs, outputs) {
        for (let i = 0; i < this._handles.length; i++) {
            const nb = inputs[i].length;
            const nbOut = (outputs && outputs[i]) ? outputs[i].length : nb;          // (the batch entry points have no output arrays: mirrored)
            const inChanged = nb !== this._channels[i], outChanged = nbOut !== this._outChannels[i];
            if (!inChanged && !outChanged) continue;
            // One corner of ola-processor.js:149-157: `outputBuffersToRetrieve` is only reallocated with the OUTPUT channels, so an output channel whose input
            // disappears keeps its last frame there and handleOutputBuffersToRetrieve goes on adding that stale frame (and shifting) every quantum.  Nobody hears it
            // (writeOutputs walks the input channels) unless the input returns before the output count changes: then the channel's pending sums are the stale frame's.
            const regained = new Map();
            if (inChanged && !outChanged) {
                for (let c = nb; c < Math.min(this._channels[i], this._outChannels[i]); c++) this._stale[i].set(c, { frame: this._staleFrame(i, c), q: 0 });
                for (let c = this._channels[i]; c < nb; c++) if (this._stale[i].has(c)) regained.set(c, this._staleAccumulator(i, c));
            }
            if (outChanged) this._stale[i].clear();                                   // allocateOutputChannels: fresh zeroed outputBuffersToRetrieve (ola-processor.js:74-85)
            if (Math.max(nb, nbOut) > this._capacity[i]) {
                // more channel slots than the handle owns: a bigger handle.  What the reference would keep across this call -- the side that did NOT change --
                // moves over (the other side starts from zeros anyway); timeCursor survives (phase-vocoder.js:31,71)
                const old = this._handles[i], oldCap = this._capacity[i], t = native.timeCursor(old);
                const keep = [];
                if (!inChanged || !outChanged) for (let c = 0; c < oldCap; c++) keep.push(native.exportState(old, c));
                native.destroy(old);
                this._capacity[i] = Math.max(nb, nbOut, 2 * oldCap);
                this._handles[i] = native.create({ fftSize: this.blockSize, hopSize: this.hopSize, maxChannels: this._capacity[i], maxHops: this._maxHops, deviceId: this.deviceId, flags: this._flags });
                native.timeCursor(this._handles[i], t);
                for (let c = 0; c < keep.length; c++) native.importState(this._handles[i], c, inChanged ? null : keep[c].hist, outChanged ? null : keep[c].acc);
            } else {
                if (inChanged) native.reset(this._handles[i], 0, this._capacity[i], STATE_HISTORY);          // allocateInputChannels: fresh zeroed input buffers
                if (outChanged) native.reset(this._handles[i], 0, this._capacity[i], STATE_ACCUMULATOR);     // allocateOutputChannels: fresh zeroed output buffers
            }
            for (const [c, acc] of regained) { native.importState(this._handles[i], c, null, acc); this._stale[i].delete(c); }
            while (this._rings[i].length < this._capacity[i]) this._rings[i].push(new Float32Array(this.blockSize));
            if (inChanged) { for (const r of this._rings[i]) r.fill(0); this._ringPos[i] = 0; }
            this._channels[i] = nb;
            this._outChannels[i] = nbOut;
        }
    }

    /** readInputs + shiftInputBuffers (ola-processor.js:89-127) on the host's copy of the input windows; one more quantum for every stale frame. */
    _afterQuantum(inputs, paused, pitchFactor) {
        const h = this.hopSize, N = this.blockSize;
        this._lastPitch = pitchFactor;
        if (N % h !== 0) return;                                // (no whole number of overlaps: the reference's buffers have no meaning there either)
        for (let i = 0; i < this._handles.length; i++) {
            const pos = this._ringPos[i], rings = this._rings[i];
            for (let c = 0; c < inputs[i].length; c++) { if (paused) rings[c].fill(0, pos, pos + h); else rings[c].set(inputs[i][c], pos); }
            this._ringPos[i] = (pos + h) % N;
            for (const rec of this._stale[i].values()) rec.q++;
        }
    }

    process(inputs, outputs, parameters) {
        this.reallocateChannelsIfNeeded(inputs, outputs);
        const pf = parameters.pitchFactor;
        const pitchFactor = pf[pf.length - 1];                  // "no automation, take last value" (phase-vocoder.js:47)
        // paused: the newest hop of EVERY input is treated as zeros (ola-processor.js:93-100)
        const paused = inputs.length > 0 && inputs[0].length > 0 && inputs[0][0].length === 0;
        for (let i = 0; i < this._handles.length; i++)
            if ((outputs[i] ? outputs[i].length : 0) < inputs[i].length)       // the reference's processOLA dereferences outputs[i][j] for every input channel (phase-vocoder.js:51)
                throw new TypeError(`outputs[${i}] has fewer channels than inputs[${i}]`);
        if (this._handles.length === 1) {
            const ins = paused ? inputs[0].map(() => PhaseVocoderProcessor._EMPTY) : inputs[0];
            native.process(this._handles[0], ins, outputs[0] || [], pitchFactor);
            this._afterQuantum(inputs, paused, pitchFactor);
            return true;                                        // ola-processor.js:170
        }
        // several inputs = several independent processors (phase-vocoder.js:49-50): launch them all, then collect them all -- every
        // kernel is in flight before the first wait, a quantum costs one exposed launch + wait instead of one per input
        const counts = new Array(this._handles.length);
        let begun = 0;
        try {
            for (; begun < this._handles.length; begun++) {
                const ins = paused ? inputs[begun].map(() => PhaseVocoderProcessor._EMPTY) : inputs[begun];
                counts[begun] = native.processBegin(this._handles[begun], ins, pitchFactor);
            }
        } catch (e) {
            // a later input failed to launch (bad block length, device error): collect the quanta already in flight so that their handles
            // do not stay "pending" for ever, then report the failure
            for (let i = 0; i < begun; i++) { try { native.processEnd(this._handles[i], outputs[i] || [], counts[i]); } catch (_) { /* the first error wins */ } }
            throw e;
        }
        for (let i = 0; i < this._handles.length; i++) native.processEnd(this._handles[i], outputs[i] || [], counts[i]);
        this._afterQuantum(inputs, paused, pitchFactor);
        return true;                                            // ola-processor.js:170
    }

    /** Throughput form (no reference counterpart): nhops process() calls of input 0 in one launch, planar [ch][nhops*hop].
     *  Needs processorOptions.maxHops >= nhops at construction; a changed channel count restarts the state (ola-processor.js:38-52). */
    processBatch(input, output, nch, nhops, pitchPerHop) {
        if (nhops > this._maxHops) throw new Error("processBatch: nhops exceeds processorOptions.maxHops");
        const fake = [Array.from({ length: nch })];
        this.reallocateChannelsIfNeeded(fake.concat(this._handles.slice(1).map((_, i) => ({ length: this._channels[i + 1] }))), null);
        return native.processBatch(this._handles[0], input, output, nch, nhops, pitchPerHop, 0, 1);
    }

    /** The same batch on a worker thread: resolves to true when the output is complete.  The handle is busy until then. */
    processBatchAsync(input, output, nch, nhops, pitchPerHop) {
        if (nhops > this._maxHops) return Promise.reject(new Error("processBatch: nhops exceeds processorOptions.maxHops"));
        const fake = [Array.from({ length: nch })];
        this.reallocateChannelsIfNeeded(fake.concat(this._handles.slice(1).map((_, i) => ({ length: this._channels[i + 1] }))), null);
        return native.processBatchAsync(this._handles[0], input, output, nch, nhops, pitchPerHop, 0, 1);
    }

    /** Everything the reference keeps for channel `ch` of input `input` between process() calls: {hist, acc, timeCursor}
     *  (inputBuffers ola-processor.js:59, outputBuffers :77, timeCursor phase-vocoder.js:31) -- checkpoint / resume / migration. */
    exportState(ch, input = 0) { return native.exportState(this._handles[input], ch); }
    importState(ch, state, input = 0) { native.importState(this._handles[input], ch, state.hist || null, state.acc || null, state.timeCursor); }

    info() { return this._handles.length ? native.info(this._handles[0]) : null; }

    close() {
        for (const h of this._handles) native.destroy(h);
        this._handles = [];
        if (this._scratch) { native.destroy(this._scratch); this._scratch = null; }
    }
}
PhaseVocoderProcessor._EMPTY = new Float32Array(0);

register("phase-vocoder-processor", PhaseVocoderProcessor);      // phase-vocoder.js:176

module.exports = { PhaseVocoderProcessor, registerProcessor: register, getProcessor: (name) => registry.get(name), native };
