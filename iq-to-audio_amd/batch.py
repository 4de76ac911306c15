"""Batches of captures that already sit in HBM: :class:`ResidentCaptureRunner`.

The per-chunk loop of the reference's ``ProcessingPipeline.run`` (processing.py:1070-1154) plus the writer's 48 kHz leg,
for whole device-resident captures with one set of settings (the ``--benchmark`` pattern, multi-file batches;
``bench.py`` is a loop over it).  Everything here is host-side sequencing of the stages in :mod:`.processing`:
streams, events, buffer slots -- no arithmetic.
"""
from __future__ import annotations

import contextlib

from ctypes import c_int32, c_int64, c_void_p

import numpy as np

from . import _dev as D
from . import _native as N
from . import dsp_plan as P
from .processing import (PRECISION_GUARD, ChannelBank, ChannelDemod, Channelizer, MixSignProbe, Resampler48k, _ChannelKernel,
                         base_precision, immutable_taps, pick_precision, probe_targets, reserve_pinned_scalars)


def _rank(precision: str) -> int:
    return _ChannelKernel.PRECISIONS.index(precision)


class ResidentCaptureRunner:
    """Ingest -> 48 kHz PCM16 for whole captures that already sit in HBM (multi-file batches with one set of
    settings; ``bench.py``): the per-chunk loop of the reference (processing.py:1070-1154) plus the writer's
    ``-ar 48000 pcm_s16le`` leg for one capture per ``submit`` -- queued without any host<->device
    synchronisation, so consecutive captures overlap on the GPU:

    * compute stream: the two mixer-sign probes (``choose_mix_sign``, processing.py:623-663) and, right behind them,
      the channelizer run *speculatively* for sign +1 -- the reference's tie-break and the common case (a signal
      at +f_off lands at DC with sign +1) -- then demodulator + writer clip + 48 kHz
      resample + PCM16;
    * egress stream: D2H of the PCM16 into pinned memory (and whatever the caller chains on ``done``),
      beside the next capture's channelizer;
    * aux stream, ``submit(..., resident=True)`` only: everything of a capture that depends on nothing but the capture
      itself -- both probes, the decoder-state reset, the float32 launch for the filter's start-up outputs -- runs
      beside the PREVIOUS capture's demodulator/resampler, so the compute stream holds nothing but channelizer,
      demodulator and resampler back to back.

    ``collect`` waits for a capture, reads the probe back and, if it chose -1 after all, re-runs that capture
    with the right sign before returning.  The same read-back carries the precision guard of the file pipeline
    (``processing.pick_precision``): the probe's channel power against the wideband level of the warm-up block; an NFM
    channel too far below it for the precision the capture ran at is re-run at the one that clears it.  The last probed
    sign and precision are the next capture's speculation.  Nothing else is cached between captures except the plans.
    SSB with the AGC on starts at the "full" precision (``processing.base_precision``).
    """

    #: submit(resident=True): demodulator + resampler of capture i on their own stream, beside the channelizer of capture i + 1
    tail_beside_next = bool(int(__import__("os").environ.get("IQA_TAIL_STREAM", "0")))
    # Output buffers in flight.  Three: the PCM16 copy of capture i (a handful of workgroups on the egress stream) is meant to
    # run beside the channelizer of capture i + 1, but a channelizer whose workgroups fill every SIMD's register file (twelve
    # waves of 160+ registers: the byte-plane kernels at 12-13 k steps) leaves it no wave slot -- the copy then completes when
    # that channelizer ends, and with two buffers the host could not queue capture i + 2 before: a bubble of one host
    # submission per capture (config 4's unit: 1.35 ms per capture for a 0.95 ms kernel).
    SLOTS = 3
    probe_behind_channelizer = bool(int(__import__("os").environ.get("IQA_PROBE_BEHIND", "1")))  # submit(resident=True): see _chain

    def __init__(self, taps: np.ndarray, *, sample_rate: float, freq_offset: float, decimation: int, fs_channel: float,
                 chunk: int, n_frames: int, demod_mode: str = "nfm", deemph_us: float = 300.0, agc_enabled: bool = True,
                 fmt: str = "s16", iq_order: str = "iq", mix_sign_override: int | None = None, precision: str | None = None,
                 precision_guard: float | None = None, slots: int | None = None, graph_streams: int = 1):
        """``precision``: the channelizer precision every capture starts at (default: by demodulator,
        ``processing.base_precision``); ``precision_guard``: see ``processing.PRECISION_GUARD`` (0 = off).
        ``slots``: captures in flight (output buffers; default 2).  ``submit`` of capture i first waits for capture
        i - slots: with captures of tens of microseconds (BASELINE config 1 replayed as hipGraphs) two in flight make every
        step a host round trip -- event wait, wake-up, graph launch -- which some hosts take 0.2 ms for; eight in flight
        keep ~0.6 ms of work queued and the step is the GPU's.
        ``graph_streams``: streams the captured steps (``submit_captured``) are replayed on, round-robin by slot.  A small
        capture's step is a chain of ten dependent kernels of a few microseconds each -- latency, not work (config 1: 80 us
        of GPU time per capture for 50 MB) -- so the chains of two or three captures side by side fill the gaps of one
        another.  Only for captures that are complete in device memory when ``submit_captured`` is called (the replay is not
        ordered behind the caller's stream)."""
        torch = D.torch_mod()
        if slots is not None:
            if slots < 2:
                raise ValueError("at least two slots")
            self.SLOTS = int(slots)
        self.taps, self.fs, self.f_off, self.d, self.fs_ch = immutable_taps(taps), float(sample_rate), float(freq_offset), int(decimation), float(fs_channel)
        self.chunk, self.n_frames, self.fmt, self.iq_order = int(chunk), int(n_frames), fmt, iq_order
        self.demod_args = dict(mode=demod_mode, deemph_us=deemph_us, agc_enabled=agc_enabled)
        self.override = mix_sign_override if mix_sign_override in (1, -1) else None
        self.base_precision = precision or base_precision(demod_mode, agc_enabled)
        self.guard = PRECISION_GUARD if precision_guard is None else float(precision_guard)
        self._guarded = bool(self.guard) and self.override is None and (demod_mode or "").lower() in ("nfm", "fm")
        self._spec_sign, self._spec_precision = 1, self.base_precision  # what the next capture is run with before its probe is read
        self.redone = dict(sign=0, precision=0)  # captures re-run in collect, by cause
        self.n_dec = -(-self.n_frames // self.d)
        self.starts = P.chunk_output_starts(self.chunk, self.d, 0, self.n_frames)
        self.rs = Resampler48k(self.fs_ch)
        self.n48 = self.rs.plan.n_out(self.n_dec)
        # One compute stream.  A second one for demod/resample beside the next channelizer was measured and dropped
        # (-6 % per capture at best while stretching the channelizer by 40 %: the small kernels take its CU slots).
        self.compute = torch.cuda.current_stream()
        self._compute_raw = int(self.compute.cuda_stream)
        self.aux = D.side_stream("aux")  # (process-wide streams, first use in this order: _dev.side_stream says why)
        self.egress = D.side_stream("egress")
        self._ring_done = None  # event behind the most recent channelizer launch (the aux stream starts from there)
        self.slots = [dict(z=D.empty(self.n_dec, "complex64"), audio=D.empty(self.n_dec, "float32"),
                           pcm_host=torch.empty(self.n48, dtype=torch.int16).pin_memory(), busy=None,
                           dem=ChannelDemod(demod_mode, self.fs_ch, deemph_us=deemph_us, agc_enabled=agc_enabled))
                      for _ in range(self.SLOTS)]
        self._next = 0
        self._graph_streams = [D.side_stream(f"graph{i}") for i in range(int(graph_streams))] if int(graph_streams) > 1 else []
        self._egress_pending = None  # ticket whose D2H has not been queued yet (see _flush_egress)
        self.egress_workgroups = 8

    def _probe(self, raw_dev, resident: bool):
        if self.override is not None:
            return None
        torch = D.torch_mod()
        warm = raw_dev[: 2 * min(self.chunk, self.n_frames)] if self.fmt != "f32" else raw_dev[: min(self.chunk, self.n_frames)]
        # record_done=False: the events that lie behind the probes are the capture's own (set in _chain) -- an event
        # record between two kernels of a stream costs ~7 us on this part
        with D.on_stream(self.aux if resident else self.compute, self.compute):
            return MixSignProbe(warm, self.fs, self.f_off, self.taps, self.d, fmt=self.fmt, iq_order=self.iq_order,
                                record_done=False, measure_level=self._guarded)

    def _kernel_for(self, sign: int):
        return lambda name: Channelizer(self.taps, sample_rate=self.fs, freq_offset=self.f_off, mix_sign=sign, decimation=self.d,
                                        fmt=self.fmt, iq_order=self.iq_order, precision=name)._kernel

    def _needed_precision(self, probe, sign: int) -> str:
        """What the precision guard asks for, given a probe that has been read (``power``, ``wideband_rms`` set)."""
        if not self._guarded or probe is None:
            return self.base_precision
        memo = self.__dict__.setdefault("_guard_memo", {}).setdefault(sign, {})
        return pick_precision(self._kernel_for(sign), self.base_precision, self.demod_args["mode"], probe.power, probe.wideband_rms,
                              self.guard, memo)

    def _chain(self, raw_dev, slot, sign: int, events=None, halo=None, resident: bool = False, probe=None, precision: str | None = None,
               make_probe=None):
        """Channelizer, demod, resample, PCM16 for one capture on the compute stream; the D2H is queued later.
        ``make_probe`` (resident captures): called right BEHIND the channelizer's launch -- the probes go to the aux stream and
        wait for nothing on the compute stream, so queueing them first only delays the launch an idle GPU is waiting for (the
        first capture behind a synchronise: ~0.1 ms of host time)."""
        torch = D.torch_mod()
        chan = Channelizer(self.taps, sample_rate=self.fs, freq_offset=self.f_off, mix_sign=sign, decimation=self.d,
                           fmt=self.fmt, iq_order=self.iq_order, precision=precision or self.base_precision)
        chan.plan_ahead()
        dem = slot["dem"]
        side = self.aux if resident else None
        with D.on_stream(self.aux if resident else self.compute, self.compute):
            dem.reset()
        dem.prepare(self.n_dec, self.starts)
        # gate: compute stream, behind this capture's probes (if they are there), in front of its channelizer -- the
        # caller's timing event when there is one (it is recorded exactly there)
        gate = None
        if events is None:
            gate = torch.cuda.Event()
            gate.record(self.compute)
        prev = self._egress_pending
        chan.process(raw_dev, out_dev=slot["z"], events=events, last_block=True, halo=halo, edge_stream=side)
        if make_probe is not None:
            probe = make_probe()
        if events is None:
            ring_done = torch.cuda.Event()
            ring_done.record(self.compute)
        else:
            gate, ring_done = events[0], events[1]
        self._ring_done = ring_done
        if prev is not None:
            self._flush_egress(gate)  # the previous capture's D2H runs beside the channelizer, not beside the probes
        aux_done = None
        if resident:
            aux_done = torch.cuda.Event()
            aux_done.record(self.aux)
            if self.tail_beside_next:
                D.side_stream("tail").wait_event(aux_done)
            else:
                self.compute.wait_event(aux_done)  # start-up outputs and decoder state are in place (long ago)
        if probe is not None:
            probe._done = [aux_done] if resident else [gate]
        tail_done = None
        if resident and self.tail_beside_next:
            # demodulator + resampler on a stream of their own: they run beside the NEXT capture's channelizer (whose
            # workgroups leave them LDS and registers on every CU, see IQA_RING_ROUNDS_MAX) instead of in front of it
            tail = D.side_stream("tail")
            tail.wait_event(ring_done)
            with D.on_stream(tail, self.compute):
                dem.process(slot["z"], self.starts, slot["audio"])
                pcm = self.rs.process(slot["audio"], want="pcm16")
                for t_ in (slot["z"], slot["audio"], pcm):
                    t_.record_stream(tail)
                tail_done = torch.cuda.Event()
                tail_done.record(tail)
        else:
            dem.process(slot["z"], self.starts, slot["audio"])
            pcm = self.rs.process(slot["audio"], want="pcm16")  # the float32 48 kHz stream is never stored
        done = torch.cuda.Event()
        # tail_done: recorded lazily (tail_event) -- the next capture's gate lies behind it anyway
        ticket = dict(chan=chan, dem=dem, pcm=pcm, done=done, tail_done=tail_done, kernel=chan._kernel.last_kernel,
                      slot=slot, egress_queued=False, resident=resident, precision=chan.precision, probe=probe)
        self._egress_pending = ticket
        return ticket

    def tail_event(self, ticket: dict):
        """Event behind the last kernel of ``ticket``'s capture (its PCM16 is complete in device memory)."""
        if ticket.get("tail_done") is None:
            ev = D.torch_mod().cuda.Event()
            ev.record(self.compute)
            ticket["tail_done"] = ev
        return ticket["tail_done"]

    def _flush_egress(self, gate=None) -> None:
        """Queue the D2H of the capture whose PCM16 is ready (or will be, behind its tail event).  Called right
        before the next capture's channelizer is launched, gated on an event behind that capture's probes, so the
        copy kernel -- whose waves sit on PCIe stores -- shares the GPU with the long HBM-bound kernel instead of the
        small latency-bound ones (measured: a probe beside the copy takes 120 us instead of 20); from ``collect``,
        ungated, for the last capture of a batch."""
        t = self._egress_pending
        if t is None or t["egress_queued"]:
            return
        torch = D.torch_mod()
        if gate is not None:
            self.egress.wait_event(gate)  # (behind the capture's last kernel too: same stream, recorded later)
            if t.get("tail_done") is not None:
                self.egress.wait_event(t["tail_done"])  # (its tail ran on a stream of its own)
        else:
            self.egress.wait_event(self.tail_event(t))
        # (called with the compute stream current: from submit/_chain and from collect)
        with D.on_stream(self.egress, self.compute):
            t["pcm"].record_stream(self.egress)
            host = t["slot"]["pcm_host"]
            # (ungated = the last capture of a batch, nothing to share the GPU with: as many workgroups as the link takes)
            N.call("iqa_trickle_copy", N.ptr(t["pcm"]), c_void_p(host.data_ptr()), c_int64(host.numel() * host.element_size()),
                   c_int32(self.egress_workgroups if gate is not None else max(self.egress_workgroups, 64)), N.stream_ptr())
            t["done"].record(self.egress)
        t["egress_queued"] = True
        self._egress_pending = None

    @staticmethod
    def padded_capture_frames(decimation: int, ntaps: int) -> tuple[int, int]:
        """(lead, slack) frames a capture buffer should carry in front of / behind the capture: ``slack`` readable
        (ignored) frames behind it make the LAST outputs interior outputs of the matrix-core channelizer (no VALU tail
        launch).  ``lead`` is 0 on purpose: a lead-in of zeros would do the same for the first outputs, but those are
        the filter's start-up transient, |z| ~ 1e-7..1e-5 -- below the fixed-point kernel's 1e-5 absolute error, so
        the NFM discriminator's phase there would be noise; the float32 VALU kernel keeps them (see ``submit``)."""
        ksteps = -(-2 * decimation // 32)
        return 0, 512 * ksteps + 34 * decimation

    def submit(self, raw_dev, events=None, enclosing=None, lead_frames: int = 0, resident: bool = False) -> dict:
        """Queue one capture (device tensor of interleaved frames, ``n_frames`` long).  Returns a ticket for ``collect``.
        ``enclosing``/``lead_frames``: ``raw_dev`` is ``enclosing[2*lead_frames : 2*(lead_frames + n_frames)]``; frames
        behind the capture may be read, ``lead_frames`` frames in front of it (if any) must be zeros
        (see ``padded_capture_frames``).  ``resident=True``: the capture's bytes are complete in device memory NOW
        (uploaded and synchronised before this call, not still being produced on a stream), so the launches that
        depend on nothing else may run on the aux stream, ahead of the compute stream.  ``events``: optional pair of
        torch events recorded directly in front of / behind the channelizer's dominant launch."""
        torch = D.torch_mod()
        if D.current_raw_stream() != self._compute_raw:
            raise RuntimeError("submit() must be called with the stream the runner was created on as the current stream")
        slot = self.slots[self._next % self.SLOTS]
        self._next += 1
        if slot["busy"] is not None:  # the slot's buffers are still owned by an earlier, uncollected capture
            self.collect(slot["busy"])
        if resident and self._ring_done is not None:
            # behind the previous channelizer: what runs on the aux stream shares the GPU with the previous capture's
            # short kernels, not with a channelizer (small kernels beside it starve and stretch it); the earlier user
            # of this slot's buffers finished before that
            self.aux.wait_event(self._ring_done)
        sign = self.override if self.override is not None else self._spec_sign
        halo = (enclosing, int(lead_frames)) if enclosing is not None else None
        if resident and self.probe_behind_channelizer:
            ticket = self._chain(raw_dev, slot, sign, events, halo, resident, None, self._spec_precision,
                                 make_probe=lambda: self._probe(raw_dev, True))
        else:
            ticket = self._chain(raw_dev, slot, sign, events, halo, resident, self._probe(raw_dev, resident), self._spec_precision)
        ticket.update(sign=sign, raw=raw_dev, halo=halo)
        slot["busy"] = ticket
        return ticket

    # ---- captured steps (hipGraph) ------------------------------------------------------------------------------------
    #
    # A capture of a few tens of MB takes the GPU less time than the host needs to queue its dozen launches through
    # Python (BASELINE config 1: channelizer 51 us of a 214 us step).  For a capture that sits in a FIXED device buffer
    # the whole step -- both probes, the edge launch, the channelizer, the three demodulator launches, the resampler and
    # the copy of the PCM16 into pinned memory -- is therefore captured once into a hipGraph (one stream, no events
    # inside) and replayed with a single host call per capture.  The probes write their powers into a pinned slot that
    # belongs to the graph; ``collect`` reads it after the replay has finished and, if it says -1, runs that capture
    # again the ordinary way with the right sign.

    def _captured_step(self, raw_dev, slot, halo):
        """Queue the whole step for the speculated sign / precision on the current (capturing) stream; returns what collect needs."""
        sign = self.override or self._spec_sign
        chan = Channelizer(self.taps, sample_rate=self.fs, freq_offset=self.f_off, mix_sign=sign, decimation=self.d,
                           fmt=self.fmt, iq_order=self.iq_order, precision=self._spec_precision)
        chan.plan_ahead()
        dem = slot["dem"]
        probe = None
        if self.override is None:
            warm = raw_dev[: 2 * min(self.chunk, self.n_frames)] if self.fmt != "f32" else raw_dev[: min(self.chunk, self.n_frames)]
            probe = MixSignProbe(warm, self.fs, self.f_off, self.taps, self.d, fmt=self.fmt, iq_order=self.iq_order, record_done=False,
                                 measure_level=self._guarded)
        dem.reset(force=True)
        dem.prepare(self.n_dec, self.starts)
        chan.process(raw_dev, out_dev=slot["z"], last_block=True, halo=halo)
        dem.process(slot["z"], self.starts, slot["audio"])
        pcm = self.rs.process(slot["audio"], want="pcm16")
        host = slot["pcm_host"]
        # (in a captured step the copy is a link of the chain, not a trickle beside the next channelizer: 64 workgroups --
        # config 1: 0.097 -> 0.0935 ms per capture)
        N.call("iqa_trickle_copy", N.ptr(pcm), c_void_p(host.data_ptr()), c_int64(host.numel() * host.element_size()),
               c_int32(max(self.egress_workgroups, 64)), N.stream_ptr())
        return dict(chan=chan, dem=dem, pcm=pcm, probe=probe, kernel=chan._kernel.last_kernel, sign=sign, precision=chan.precision)

    def submit_captured(self, raw_dev, enclosing=None, lead_frames: int = 0) -> dict:
        """``submit`` for a capture in a fixed buffer: the first call per (buffer, slot) runs the step once the ordinary way
        (plans, tap uploads, pinned slots come into being), captures it into a graph and replays it; later calls replay.
        Returns a ticket for ``collect``."""
        torch = D.torch_mod()
        if D.current_raw_stream() != self._compute_raw:
            raise RuntimeError("submit_captured() must be called with the stream the runner was created on as the current stream")
        index = self._next % self.SLOTS
        slot = self.slots[index]
        if slot["busy"] is not None:
            self.collect(slot["busy"])
        halo = (enclosing, int(lead_frames)) if enclosing is not None else None
        def graph_key():  # (a graph holds the speculation it was captured with)
            return (int(raw_dev.data_ptr()), index, None if enclosing is None else int(enclosing.data_ptr()), int(lead_frames),
                    self._spec_sign, self._spec_precision)

        key = graph_key()
        graphs = self.__dict__.setdefault("_graphs", {})
        entry = graphs.get(key)
        if entry is None:
            # once the ordinary way, in this very slot: plans, tap uploads and the pinned probe slot exist afterwards
            # (and the capture's own probe has set the speculation the graph is captured with)
            self.collect(self.submit(raw_dev, enclosing=enclosing, lead_frames=lead_frames))
            key = graph_key()
            torch.cuda.synchronize()
            reserve_pinned_scalars(1)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                parts = self._captured_step(raw_dev, slot, halo)
            entry = graphs[key] = dict(graph=g, **parts)
        else:
            self._next += 1
        entry["dem"].chunk_sumsq = entry["dem"].chunk_sumsq[-1:]  # (a replay re-runs the same demodulator call)
        done = entry.get("done_event")  # (one event per graph: its slot's previous capture was collected above)
        if done is None:
            done = entry["done_event"] = torch.cuda.Event()
        if self._graph_streams:  # this slot's stream: the chains of consecutive captures run side by side
            side = self._graph_streams[index % len(self._graph_streams)]
            with torch.cuda.stream(side):
                entry["graph"].replay()
                done.record(side)
        else:
            entry["graph"].replay()
            done.record()
        ticket = dict(chan=entry["chan"], dem=entry["dem"], pcm=entry["pcm"], done=done, tail_done=done, kernel=entry["kernel"],
                      slot=slot, egress_queued=True, resident=False, probe=None, graph_probe=entry["probe"],
                      sign=entry["sign"], precision=entry["precision"], raw=raw_dev, halo=halo)
        slot["busy"] = ticket
        return ticket

    def submit_captured_batch(self, captures: list) -> list:
        """``submit_captured`` for up to ``SLOTS`` captures at once -- ``captures``: ``(raw_dev, enclosing, lead_frames)`` each,
        in fixed buffers -- as ONE graph: one host call and one graph launch (≈13 us on this part) for all of them.
        Returns one ticket per capture."""
        torch = D.torch_mod()
        if D.current_raw_stream() != self._compute_raw:
            raise RuntimeError("submit_captured_batch() must be called with the stream the runner was created on as the current stream")
        if not 1 <= len(captures) <= self.SLOTS:
            raise ValueError(f"a batch holds 1..{self.SLOTS} captures")
        for slot in self.slots[: len(captures)]:
            if slot["busy"] is not None:
                self.collect(slot["busy"])
        halos = [(enc, int(lead)) if enc is not None else None for _, enc, lead in captures]
        def graph_key():
            return ("batch", self._spec_sign, self._spec_precision) + tuple((int(raw.data_ptr()), None if enc is None else int(enc.data_ptr()), int(lead))
                                                                            for raw, enc, lead in captures)

        key = graph_key()
        graphs = self.__dict__.setdefault("_graphs", {})
        entry = graphs.get(key)
        if entry is None:
            self._next = 0
            for raw, enc, lead in captures:  # once the ordinary way, each in its slot
                self.collect(self.submit(raw, enclosing=enc, lead_frames=lead))
            key = graph_key()
            torch.cuda.synchronize()
            reserve_pinned_scalars(len(captures))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                parts = [self._captured_step(raw, slot, halo) for (raw, _, _), slot, halo in zip(captures, self.slots, halos)]
            entry = graphs[key] = dict(graph=g, parts=parts)
        for p_ in entry["parts"]:
            p_["dem"].chunk_sumsq = p_["dem"].chunk_sumsq[-1:]
        entry["graph"].replay()
        done = torch.cuda.Event()
        done.record()
        tickets = []
        for (raw, _, _), slot, halo, p_ in zip(captures, self.slots, halos, entry["parts"]):
            t = dict(chan=p_["chan"], dem=p_["dem"], pcm=p_["pcm"], done=done, tail_done=done, kernel=p_["kernel"], slot=slot,
                     egress_queued=True, resident=False, probe=None, graph_probe=p_["probe"], sign=p_["sign"], precision=p_["precision"],
                     raw=raw, halo=halo)
            slot["busy"] = t
            tickets.append(t)
        return tickets

    def collect(self, ticket: dict) -> dict:
        """Wait for a submitted capture.  Returns {"pcm_host", "sign", "demod" (``.peak``, ``.chunk_rms_dbfs()``),
        "audio", "z", "kernel"}; the buffers belong to the runner and are reused ``SLOTS`` submits later."""
        slot = ticket["slot"]
        if ticket.get("collected"):
            return ticket["result"]
        if ticket.get("graph_probe") is not None:  # a replayed step: its probes' powers sit in the graph's pinned slot
            ticket["done"].synchronize()
            sign = ticket["graph_probe"].peek()
            need = self._needed_precision(ticket["graph_probe"], sign)
            self._spec_sign, self._spec_precision = sign, need
            if sign != ticket["sign"] or _rank(need) > _rank(ticket["precision"]):
                # the captured step assumed another sign / a coarser precision: this capture again, the ordinary way
                self.replays_redone = getattr(self, "replays_redone", 0) + 1
                self.redone["sign" if sign != ticket["sign"] else "precision"] += 1
                redo = self._chain(ticket["raw"], slot, sign, None, ticket.get("halo"), False, precision=need)
                self._flush_egress()
                ticket.update(redo, sign=sign, probe=None, raw=ticket["raw"], graph_probe=None)
        if D.current_raw_stream() != self._compute_raw:  # (it may queue the D2H, or the whole capture again)
            raise RuntimeError("collect() must be called with the stream the runner was created on as the current stream")
        sign = ticket["sign"]
        if self._egress_pending is ticket:
            self._flush_egress()
        if ticket["probe"] is not None:
            sign = ticket["probe"].result()
            need = self._needed_precision(ticket["probe"], sign)
            self._spec_sign, self._spec_precision = sign, need
            if sign != ticket["sign"] or _rank(need) > _rank(ticket["precision"]):
                # the speculation was wrong: this capture again, with the sign the probe chose / at the precision the guard asks for
                self.redone["sign" if sign != ticket["sign"] else "precision"] += 1
                ticket["done"].synchronize()
                probe, raw = ticket["probe"], ticket["raw"]
                redo = self._chain(raw, slot, sign, None, ticket.get("halo"), ticket.get("resident", False), precision=need)
                self._flush_egress()
                ticket.update(redo, sign=sign, probe=probe, raw=raw)
        ticket["done"].synchronize()
        ticket["collected"] = True
        ticket["raw"] = ticket["pcm"] = None  # back to the allocator: the next capture reuses them
        if slot["busy"] is ticket:
            slot["busy"] = None
        ticket["result"] = dict(pcm_host=slot["pcm_host"], sign=int(sign), audio=slot["audio"], z=slot["z"],
                                kernel=ticket["kernel"], demod=ticket["dem"], done=ticket["done"], precision=ticket["precision"])
        return ticket["result"]


class ResidentBankRunner:
    """Several ``--ft`` targets of whole captures that already sit in HBM (BASELINE config 3; config 5's per-GPU unit).

    The reference runs one pipeline per target over the same file (cli.py:683-710).  Here one ``submit`` queues, for one
    resident capture: every target's two mixer-sign probes (``choose_mix_sign``, processing.py:623-663), ONE pass of the
    channelizer over the capture for all targets (:class:`processing.ChannelBank`, run speculatively for sign +1 like
    :class:`ResidentCaptureRunner`), then per target the fused demodulator + writer clip, the 48 kHz resampler with PCM16
    output and the copy of that PCM16 into pinned host memory -- all on the caller's stream, no host<->device
    synchronisation.  ``collect`` waits, reads the probes back and re-runs the targets whose probe chose the other sign
    or whose level asks for a finer precision (the precision guard, ``processing.pick_precision``); what a target's last
    probe said is the next capture's speculation.  SSB targets with the AGC on run at "full" precision (their own chained
    passes behind the shared one, ``processing.base_precision``).
    """

    SLOTS = 3  # output buffers in flight: with the tails of capture i finishing somewhere inside the pass of capture i + 1, a
               # third slot lets the host queue capture i + 2 without waiting for them
    overlap_tails = True  # the per-target chains of capture i beside the channelizer pass of capture i + 1
    # Where a capture's float32 edge launches and combine launches go: False = the caller's stream, between the passes (they
    # are short); True = the tails' stream, "own" = a third stream.  On a side stream they kept the caller's stream free for
    # the passes while the previous capture's chains found room BESIDE a pass; behind a pass of twelve waves x 160+
    # registers per workgroup the chains finish late, and everything queued behind them -- these launches, and with them the
    # next pass -- waited: config 3 measured 13.6 ms per capture (= everything on one stream) against 13.2-13.3 with them on
    # the caller's stream (profiles/r03_c3_tail_modes.txt).
    edges_on_side = False
    #: the mixer-sign probes of a capture on the tails' stream (read at ``collect`` only) instead of the caller's, in front of its pass
    probes_on_side = bool(int(__import__("os").environ.get("IQA_PROBES_ON_SIDE", "0")))
    #: streams the targets' chains of one capture are spread over (target i on stream i % tail_streams).  One: the chains
    #: follow one another (each is a handful of small dependent kernels, latency-bound).
    tail_streams = int(__import__("os").environ.get("IQA_TAIL_STREAMS", "1"))
    #: the next capture's pass waits for this capture's chains (they then have the whole part to themselves -- for passes
    #: whose workgroups leave no registers for another kernel's waves beside them)
    pass_waits_for_tails = bool(int(__import__("os").environ.get("IQA_PASS_WAITS_FOR_TAILS", "0")))

    def __init__(self, targets: list, *, sample_rate: float, n_frames: int, chunk_size: int = 1_048_576,
                 fs_ch_target: float = 96_000.0, fmt: str = "s16", iq_order: str = "iq", precision_guard: float | None = None):
        """``targets``: dicts with ``freq_offset``, and optionally ``bandwidth`` (12 500), ``demod_mode`` ("nfm"),
        ``deemph_us`` (300), ``agc_enabled`` (True), ``mix_sign`` (None = probe), ``precision`` (None = by demodulator)."""
        torch = D.torch_mod()
        if not targets:
            raise ValueError("at least one target is required")
        self.fs, self.n_frames, self.fmt, self.iq_order = float(sample_rate), int(n_frames), fmt, iq_order
        self.d, self.fs_ch = P.choose_decimation(self.fs, fs_ch_target)
        self.chunk = P.tune_chunk_size(self.fs, chunk_size)
        self.n_dec = -(-self.n_frames // self.d)
        self.starts = P.chunk_output_starts(self.chunk, self.d, 0, self.n_frames)
        self.rs = Resampler48k(self.fs_ch)
        self.n48 = self.rs.plan.n_out(self.n_dec)
        self.targets = []
        for t in targets:
            spec = dict(bandwidth=12_500.0, demod_mode="nfm", deemph_us=300.0, agc_enabled=True, mix_sign=None, precision=None)
            spec.update(t)
            spec["taps"] = immutable_taps(P.design_channel_filter(self.fs, spec["bandwidth"], self.d))
            spec["base_precision"] = spec["precision"] or base_precision(spec["demod_mode"], spec["agc_enabled"])
            self.targets.append(spec)
        self.guard = PRECISION_GUARD if precision_guard is None else float(precision_guard)
        self._spec_sign = [1] * len(self.targets)
        self._spec_precision = [s["base_precision"] for s in self.targets]
        self.redone = dict(sign=0, precision=0)
        self.slots = []
        for _ in range(self.SLOTS):
            per = [dict(z=D.empty(self.n_dec, "complex64"), audio=D.empty(self.n_dec, "float32"),
                        pcm_host=torch.empty(self.n48, dtype=torch.int16).pin_memory(),
                        dem=ChannelDemod(s["demod_mode"], self.fs_ch, deemph_us=s["deemph_us"], agc_enabled=s["agc_enabled"]))
                   for s in self.targets]
            # (two powers per target + the wideband level of the warm-up block)
            self.slots.append(dict(per=per, busy=None, probe_host=torch.zeros(2 * len(self.targets) + 1, dtype=torch.float64).pin_memory()))
        self._next = 0

    def _channelizer(self, spec, sign: int, precision: str | None = None) -> Channelizer:
        return Channelizer(spec["taps"], sample_rate=self.fs, freq_offset=spec["freq_offset"], mix_sign=sign, decimation=self.d,
                           fmt=self.fmt, iq_order=self.iq_order, precision=precision or spec["base_precision"])

    def _guarded(self, spec) -> bool:
        return bool(self.guard) and spec["mix_sign"] not in (1, -1) and (spec["demod_mode"] or "").lower() in ("nfm", "fm")

    def _needed_precision(self, spec, probe, sign: int) -> str:
        if probe is None or not self._guarded(spec):
            return spec["base_precision"]
        memo = spec.setdefault("_guard_memo", {}).setdefault(sign, {})
        return pick_precision(lambda name: self._channelizer(spec, sign, name)._kernel, spec["base_precision"], spec["demod_mode"],
                              probe.power, probe.wideband_rms, self.guard, memo)

    def _finish_target(self, per, raw_unused=None) -> None:
        """Demodulator + writer clip, 48 kHz PCM16, copy to the host: for one target's z."""
        dem = per["dem"]
        dem.reset()
        dem.process(per["z"], self.starts, per["audio"])
        pcm = self.rs.process(per["audio"], want="pcm16")
        per["pcm_host"].copy_(pcm, non_blocking=True)
        per["pcm"] = pcm  # (the device copy: what an N-GPU job hands to the RCCL gather; valid until this slot's next capture)

    def submit(self, raw_dev, enclosing=None, lead_frames: int = 0, events=None) -> dict:
        """Queue one capture; ``events``: optional pair of torch events recorded around the channelizer's pass."""
        torch = D.torch_mod()
        slot = self.slots[self._next % self.SLOTS]
        self._next += 1
        if slot["busy"] is not None:
            self.collect(slot["busy"])
        warm = raw_dev[: 2 * min(self.chunk, self.n_frames)] if self.fmt != "f32" else raw_dev[: min(self.chunk, self.n_frames)]
        main = torch.cuda.current_stream()
        side = D.side_stream("tail") if self.overlap_tails else None
        if side is not None:  # the capture is resident when submit is called: what depends on it alone may start now
            arrived = torch.cuda.Event()
            arrived.record(main)
            side.wait_event(arrived)
        # (the probes stay on the caller's stream: as ring-kernel launches they want most of a CU's LDS and would sit behind
        # the running pass until it ends, one per pass boundary; through the float32 kernel -- MixSignProbe(matrix_cores=
        # False) -- they fit beside it but the long filters' probes then take longer than the pass has room for: 10.4
        # against 9.4 ms per capture at config 3)
        probe_ctx = D.on_stream(side, main) if (self.probes_on_side and side is not None) else contextlib.nullcontext()
        with probe_ctx:
            probes = self._queue_probes(warm, slot)
        signs = [s["mix_sign"] if s["mix_sign"] in (1, -1) else sp for s, sp in zip(self.targets, self._spec_sign)]
        precisions = list(self._spec_precision)
        chans = [self._channelizer(s, sg, pr) for s, sg, pr in zip(self.targets, signs, precisions)]
        for c in chans:
            c.plan_ahead()
        halo = (enclosing, int(lead_frames)) if enclosing is not None else None
        return self._submit_rest(raw_dev, slot, probes, signs, precisions, chans, halo, events, main, side, arrived if side is not None else None)

    def _queue_probes(self, warm, slot) -> list:
        todo = [i for i, s in enumerate(self.targets) if s["mix_sign"] not in (1, -1)]
        grouped = probe_targets(warm, self.fs, [(self.targets[i]["freq_offset"], self.targets[i]["taps"]) for i in todo], self.d,
                                fmt=self.fmt, iq_order=self.iq_order, host=slot["probe_host"]) if len(todo) > 1 else None
        if grouped is not None:  # every target's two probes as channels of one bank over the snippet
            probes = [None] * len(self.targets)
            for i, pr in zip(todo, grouped):
                probes[i] = pr
        else:
            probes = [None if s["mix_sign"] in (1, -1) else
                      MixSignProbe(warm, self.fs, s["freq_offset"], s["taps"], self.d, fmt=self.fmt, iq_order=self.iq_order,
                                   measure_level=self._guarded(s))
                      for s in self.targets]
        return probes

    def _submit_rest(self, raw_dev, slot, probes, signs, precisions, chans, halo, events, main, side, arrived) -> dict:
        torch = D.torch_mod()
        if self.pass_waits_for_tails:
            for ev in self.__dict__.pop("_tails_done", []):
                main.wait_event(ev)
        if events:
            events[0].record()
        bank = ChannelBank(chans)
        # edges_on_side: True = the tails' stream (one queue: edges, combines and chains of the captures one after the other);
        # "own" = a stream of their own -- the chains of capture i - 1, which find little room beside a pass whose workgroups fill
        # the register files and finish late, then do not stand between pass i and its combine launches
        own = self.edges_on_side in ("own", "edges")
        edge = None if not self.edges_on_side or side is None else (D.side_stream("edge") if own else side)
        if edge is not None and edge is not side:
            edge.wait_event(arrived)
        bank.combines_on_edge_stream = self.edges_on_side != "edges"  # "edges": only the edge launches leave the caller's stream
        bank.process(raw_dev, outs=[p["z"] for p in slot["per"]], last_block=True, halo=halo, edge_stream=edge)
        if edge is not None and edge is not side:
            combined = torch.cuda.Event()
            combined.record(edge)
            side.wait_event(combined)
        if events:
            events[1].record()
        self.last_bank_launches = getattr(bank, "launches", None) or [bank.last_launch]  # (one entry per shared launch of the capture)
        # The targets' demodulator / resampler / copy chains run on ONE side stream behind the channelizer pass (and behind
        # this capture's probes and float32 edge launches, queued there above), so the next capture's channelizer pass
        # (other slot, caller's stream) does not wait for them.  The pass is
        # matrix-bound and leaves 16 CUs idle (240 workgroups): the small kernels find room there and beside it.  (One
        # stream per TARGET was measured and dropped: 14.2 against 11.3 ms per capture.)
        if side is not None:
            after_pass = torch.cuda.Event()
            after_pass.record(main)
            side.wait_event(after_pass)
            sides = [side] + [D.side_stream(f"tail{k}") for k in range(1, max(1, min(self.tail_streams, len(slot["per"]))))]
            for extra in sides[1:]:  # (behind the pass and behind what the first side stream had queued before it: probes, edges, combines)
                behind_side = torch.cuda.Event()
                behind_side.record(side)
                extra.wait_event(behind_side)
            for k, per in enumerate(slot["per"]):
                st = sides[k % len(sides)]
                with D.on_stream(st, main):
                    self._finish_target(per)
                    for key in ("z", "audio"):
                        per[key].record_stream(st)
            done = torch.cuda.Event()
            for extra in sides[1:]:
                joined = torch.cuda.Event()
                joined.record(extra)
                side.wait_event(joined)
            done.record(side)
            if self.pass_waits_for_tails:
                self._tails_done = [done]
        else:
            for per in slot["per"]:
                self._finish_target(per)
            done = torch.cuda.Event()
            done.record()
        ticket = dict(slot=slot, probes=probes, signs=signs, precisions=[c.precision for c in chans], raw=raw_dev, halo=halo, done=done,
                      launch=bank.last_launch, kernel=chans[0]._kernel.last_kernel, collected=False)
        slot["busy"] = ticket
        return ticket

    def collect(self, ticket: dict) -> list:
        """Wait for a capture.  Returns one dict per target: {"pcm_host", "audio", "z", "sign", "demod"}; the buffers
        belong to the runner and are reused ``SLOTS`` submits later."""
        if ticket["collected"]:
            return ticket["result"]
        slot = ticket["slot"]
        ticket["done"].synchronize()
        redo = False
        for i, (probe, per, spec) in enumerate(zip(ticket["probes"], slot["per"], self.targets)):
            if probe is None:
                continue
            sign = probe.result()
            need = self._needed_precision(spec, probe, sign)
            self._spec_sign[i], self._spec_precision[i] = sign, need
            if sign != ticket["signs"][i] or _rank(need) > _rank(ticket["precisions"][i]):
                # the speculation was wrong for this target (sign, or a level that asks for a finer precision): its channel again, alone
                self.redone["sign" if sign != ticket["signs"][i] else "precision"] += 1
                ticket["signs"][i] = sign
                chan = self._channelizer(spec, sign, need)
                ticket["precisions"][i] = chan.precision
                chan.process(ticket["raw"], out_dev=per["z"], last_block=True, halo=ticket["halo"])
                self._finish_target(per)
                redo = True
        if redo:
            D.torch_mod().cuda.current_stream().synchronize()
        ticket["collected"] = True
        ticket["raw"] = None
        if slot["busy"] is ticket:
            slot["busy"] = None
        ticket["result"] = [dict(pcm_host=per["pcm_host"], pcm=per.get("pcm"), audio=per["audio"], z=per["z"], sign=int(sg), demod=per["dem"], precision=pr)
                            for per, sg, pr in zip(slot["per"], ticket["signs"], ticket["precisions"])]
        return ticket["result"]


def demodulate_sharded(targets: list, *, sample_rate: float, n_frames: int, axis: str, capture=None, captures=None,
                       chunk_size: int = 1_048_576, fmt: str = "s16", iq_order: str = "iq"):
    """The N-GPU form of :class:`ResidentBankRunner` (one process per GPU under ``torch.distributed.run``; SURVEY.md
    section 8(e)), on either axis:

    * ``axis="channels"`` -- **channels of one capture** (BASELINE config 5): ``capture`` is the 1-D tensor of interleaved
      values on rank 0 (``None`` on the other ranks); it is replicated with ONE RCCL broadcast, every rank extracts its
      contiguous share of ``targets`` in one pass over it (a bank), rank 0 receives every target's 48 kHz PCM16.
      Returns ``({target index: int16 ndarray}, peak)`` on rank 0.
    * ``axis="captures"`` -- **independent captures** (config 4): ``captures`` has one entry per capture, a 1-D tensor or a
      zero-argument callable that loads one (only the owning rank calls it); every rank runs all ``targets`` on its share.
      Returns ``({capture index: int16 ndarray of shape (len(targets), n48)}, peak)`` on rank 0.

    ``(None, peak)`` on the other ranks; without a process group it is the single-GPU path.  The only collectives are
    the broadcast (channel axis) and the final gather of the audio."""
    from . import dist as DS

    torch = D.torch_mod()
    if axis not in ("channels", "captures"):
        raise ValueError("axis must be 'channels' or 'captures'")

    def run_bank(specs, raw_dev):
        runner = ResidentBankRunner(specs, sample_rate=sample_rate, n_frames=n_frames, chunk_size=chunk_size, fmt=fmt, iq_order=iq_order)
        res = runner.collect(runner.submit(raw_dev))
        return [(r["pcm"].clone(), r["demod"].peak) for r in res]  # (device PCM16: the gather is RCCL's)

    if axis == "channels":
        dtype = {"s16": torch.int16, "u8": torch.uint8, "f32": torch.float32}[fmt]
        shared = dict(tensor=capture, numel=2 * n_frames, dtype=dtype, device=D.device())
        return DS.run_sharded(list(targets), run_bank, shared=shared)

    def every_target_of(mine, _):
        out = []
        for unit in mine:
            raw = (unit() if callable(unit) else unit).to(D.device())
            res = run_bank(list(targets), raw)
            out.append((torch.cat([a for a, _ in res]), max(p for _, p in res)))
        return out

    got, peak = DS.run_sharded(list(captures), every_target_of)
    if got is not None:
        got = {k: v.reshape(len(targets), -1) for k, v in got.items()}
    return got, peak
