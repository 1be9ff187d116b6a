"""Python mirror of demod::DemodTask (src/demod.rs:25-120) over the C ABI.

Same wiring as the reference: a reader channel of u8 I/Q chunks in, baseband chunks out on the
receiver channel, signal power to the hub every 4th chunk.  Channels are anything with get()/put()
(queue.Queue); `None` on the reader channel ends run() (the reference loops until the process dies).
All arithmetic happens in libp25fe.so on the GPU.
"""
from .consts import BUF_BYTES
from .frontend import FrontEnd


class HubEvent:
    """hub::HubEvent::UpdateSignalPower (src/hub.rs:455)."""

    def __init__(self, kind, value):
        self.kind, self.value = kind, value

    @staticmethod
    def UpdateSignalPower(p):
        return HubEvent("UpdateSignalPower", p)


class RecvEvent:
    """recv::RecvEvent (src/recv.rs:23-30)."""

    def __init__(self, kind, value=None):
        self.kind, self.value = kind, value

    @staticmethod
    def Baseband(samples):
        return RecvEvent("Baseband", samples)

    @staticmethod
    def SetControlFreq(freq):
        return RecvEvent("SetControlFreq", freq)

    @staticmethod
    def ResetStats():
        return RecvEvent("ResetStats")


class Throttler:
    """throttle::Throttler: run the closure every n-th call (src/demod.rs:67, 95)."""

    def __init__(self, n):
        self.n, self.i = n, 0

    def throttle(self):
        self.i += 1
        if self.i == self.n:
            self.i = 0
            return True
        return False


class DemodTask:
    def __init__(self, reader, hub, chan, frontend=None):
        """DemodTask::new (src/demod.rs:44-59): decimation 5, average 10, FM 5 kHz @ 48 kHz are in the library."""
        self.reader, self.hub, self.chan = reader, hub, chan
        self.fe = frontend or FrontEnd()

    def run(self):
        """DemodTask::run (src/demod.rs:62-119)."""
        notifier = Throttler(4)                                   # :67
        while True:
            data = self.reader.get()                              # :70
            if data is None:
                return
            assert len(data) % 2 == 0 and len(data) <= BUF_BYTES * 64
            want = notifier.throttle()                            # :95
            if want:
                bb, power = self.fe.demod_u8(data, want_power=True)     # :74-93, 97, 109-114
                self.hub.put(HubEvent.UpdateSignalPower(power))   # :99
            else:
                bb = self.fe.demod_u8(data)
            self.chan.put(RecvEvent.Baseband(bb))                 # :116
