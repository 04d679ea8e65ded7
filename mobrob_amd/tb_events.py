"""Scalar summaries in TensorBoard's event-file format, without the tensorboard package.

The reference passes `tensorboard_log=<DATA_DIR>/policies/tmp/<env>-ppo/tensorboard` to SB3's PPO whenever tensorboard is
importable (/root/reference/src/mobrob/rl_control/ppo.py:50-56); SB3 then writes one run directory per `learn()` call,
`<tensorboard_log>/<tb_log_name>_<n>/events.out.tfevents.*`, with the scalars of every logged iteration at
step = num_timesteps.  This module writes the same files so that `tensorboard --logdir` shows a run of this engine next to a
run of the reference:

* record framing (TFRecord): u64 length | masked crc32c(length) | payload | masked crc32c(payload);
* payload: a serialized `tensorflow.Event` -- wall_time (field 1, double), step (2, int64), file_version (3, string; first
  record, "brain.Event:2") or summary (5) = repeated Summary.Value {tag (1, string), simple_value (2, float)}.

Only the fields above are produced; the protobuf wire encoding of those few fields is written out by hand.
"""
import os
import re
import socket
import struct
import time

_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1   # Castagnoli polynomial, reflected
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data: bytes) -> int:
    t, c = _crc_table(), 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _len_field(field: int, payload: bytes) -> bytes:
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def encode_event(wall_time: float, step: int = 0, scalars=None, file_version: str = None) -> bytes:
    ev = _varint(1 << 3 | 1) + struct.pack("<d", float(wall_time)) + _varint(2 << 3 | 0) + _varint(int(step))
    if file_version is not None:
        ev += _len_field(3, file_version.encode())
    if scalars:
        summary = b"".join(
            _len_field(1, _len_field(1, str(tag).encode()) + _varint(2 << 3 | 5) + struct.pack("<f", float(v)))
            for tag, v in scalars)
        ev += _len_field(5, summary)
    return ev


def frame(payload: bytes) -> bytes:
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", masked_crc(head)) + payload + struct.pack("<I", masked_crc(payload))


def next_run_dir(log_root: str, name: str, continue_latest: bool = False) -> str:
    """SB3's `get_latest_run_id` rule: `<name>_<k>` with k = 1 + the largest existing suffix (the largest itself when a
    run is continued with reset_num_timesteps=False)."""
    latest = 0
    if os.path.isdir(log_root):
        for d in os.listdir(log_root):
            m = re.fullmatch(re.escape(name) + r"_(\d+)", d)
            if m and os.path.isdir(os.path.join(log_root, d)):
                latest = max(latest, int(m.group(1)))
    if continue_latest and latest > 0:
        return os.path.join(log_root, f"{name}_{latest}")
    return os.path.join(log_root, f"{name}_{latest + 1}")


class EventFileWriter:
    def __init__(self, run_dir: str):
        os.makedirs(run_dir, exist_ok=True)
        now = time.time()
        self.path = os.path.join(run_dir, f"events.out.tfevents.{int(now)}.{socket.gethostname()}.{os.getpid()}.0")
        self._f = open(self.path, "ab")
        self._f.write(frame(encode_event(now, 0, file_version="brain.Event:2")))
        self._f.flush()

    def add_scalars(self, scalars, step: int):
        """scalars: iterable of (tag, number); written as ONE event at `step` and flushed."""
        rows = [(k, v) for k, v in scalars if isinstance(v, (int, float)) and v == v]   # SB3's writer gets no NaN either way
        if rows:
            self._f.write(frame(encode_event(time.time(), step, rows)))
            self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None


def read_events(path: str):
    """Parse a file back (tests, and a quick look without TensorBoard): list of dicts with wall_time / step /
    file_version / scalars; every CRC is checked."""
    def varint(b, i):
        n = s = 0
        while True:
            n |= (b[i] & 0x7F) << s
            s += 7
            i += 1
            if not b[i - 1] & 0x80:
                return n, i

    def fields(b):
        i = 0
        while i < len(b):
            key, i = varint(b, i)
            f, wt = key >> 3, key & 7
            if wt == 0:
                v, i = varint(b, i)
            elif wt == 1:
                v, i = b[i:i + 8], i + 8
            elif wt == 5:
                v, i = b[i:i + 4], i + 4
            elif wt == 2:
                n, i = varint(b, i)
                v, i = b[i:i + n], i + n
            else:
                raise ValueError(f"wire type {wt}")
            yield f, v

    out = []
    with open(path, "rb") as fh:
        data = fh.read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (n,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[i + 8:i + 12])[0] != masked_crc(head):
            raise ValueError("length CRC mismatch")
        payload = data[i + 12:i + 12 + n]
        if struct.unpack("<I", data[i + 12 + n:i + 16 + n])[0] != masked_crc(payload):
            raise ValueError("payload CRC mismatch")
        i += 16 + n
        ev = dict(scalars={})
        for f, v in fields(payload):
            if f == 1:
                ev["wall_time"] = struct.unpack("<d", v)[0]
            elif f == 2:
                ev["step"] = v
            elif f == 3:
                ev["file_version"] = v.decode()
            elif f == 5:
                for f2, val in fields(v):
                    if f2 == 1:
                        tag, num = None, None
                        for f3, x in fields(val):
                            if f3 == 1:
                                tag = x.decode()
                            elif f3 == 2:
                                num = struct.unpack("<f", x)[0]
                        ev["scalars"][tag] = num
        out.append(ev)
    return out
