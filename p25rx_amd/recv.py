"""Python mirror of the sample path of recv::RecvTask (src/recv.rs:140-167, 204-210) over the C ABI.

The reference feeds every baseband sample to p25::MessageReceiver::feed and reacts to decoded
packets; this build replaces the front half of feed() (sync, timing, slicer) and hands the dibits
and frame-sync positions to `sink` -- where the unchanged downstream p25 layers (NID, FEC, trunking)
would attach.  Policy, talkgroups, hub and audio are out of scope.
"""
from .demod import HubEvent, Throttler
from .frontend import FrontEnd


class RecvTask:
    def __init__(self, events, sink, frontend=None, hub=None):
        self.events, self.sink, self.hub = events, sink, hub
        self.fe = frontend or FrontEnd()
        self.stats = {"dibits": 0, "syncs": 0}                     # what exists of p25::stats::Stats on this path

    def set_freq(self, freq):
        """RecvTask::set_freq (src/recv.rs:127-137): after retuning, drop symbol lock (msg.resync(), :136)."""
        self.fe.resync()

    def run(self, cb=lambda samples: None):
        """RecvTask::run (src/recv.rs:140-167).  `cb` sees every baseband chunk (:152, the -w dump hook)."""
        stats_notifier = Throttler(16)                            # :141
        while True:
            ev = self.events.get()                                 # :144
            if ev is None:
                return
            if ev.kind == "Baseband":                              # :145
                dibits, sync_pos, sync_dibit = self.fe.slice(ev.value)    # :148-150 -> msg.feed(s), front half
                self.stats["dibits"] += len(dibits)
                self.stats["syncs"] += len(sync_pos)
                self.sink(dibits, sync_pos, sync_dibit)
                cb(ev.value)                                       # :152
            elif ev.kind == "SetControlFreq":                      # :158 -> set_control_freq -> switch_control -> set_freq
                self.set_freq(ev.value)
            elif ev.kind == "ResetStats":                          # :159
                self.stats = {"dibits": 0, "syncs": 0}
            if stats_notifier.throttle() and self.hub is not None:     # :162-165: after EVERY 16th event of any kind
                self.hub.put(HubEvent("UpdateStats", dict(self.stats)))
