"""The slice of spinup.utils.logx.EpochLogger that example/dsac.py:153-177 uses (spinup is a third-party
dependency, absent here and unpinned; restated from its published behaviour): `EpochLogger(output_dir=...,
output_fname='progress.txt', exp_name=None)`, `save_config(dict)` -> config.json, `log_tabular(key, val)`,
`dump_tabular()` -> one tab-separated row per call in <output_dir>/progress.txt (header on the first dump) and an
aligned table on stdout."""
import json
import os
import time


def setup_logger_kwargs(exp_name, seed=None, data_dir=None):
    """spinup.utils.run_utils.setup_logger_kwargs: <data_dir>/<exp_name>/<exp_name>_s<seed>."""
    data_dir = data_dir or os.path.join(os.getcwd(), "data")
    sub = exp_name if seed is None else os.path.join(exp_name, "%s_s%d" % (exp_name, seed))
    return dict(output_dir=os.path.join(data_dir, sub), exp_name=exp_name)


class EpochLogger:
    def __init__(self, output_dir=None, output_fname="progress.txt", exp_name=None, quiet=False):
        self.output_dir = output_dir or "/tmp/experiments/%i" % int(time.time())
        os.makedirs(self.output_dir, exist_ok=True)
        self.output_file = open(os.path.join(self.output_dir, output_fname), "w")
        self.first_row, self.log_headers, self.log_current_row = True, [], {}
        self.exp_name, self.quiet = exp_name, quiet

    def save_config(self, config):
        def conv(o):
            try:
                json.dumps(o)
                return o
            except TypeError:
                if isinstance(o, dict):
                    return {str(k): conv(v) for k, v in o.items()}
                if isinstance(o, (list, tuple)):
                    return [conv(v) for v in o]
                return str(o)
        cfg = conv(dict(config))
        if self.exp_name is not None:
            cfg["exp_name"] = self.exp_name
        with open(os.path.join(self.output_dir, "config.json"), "w") as f:
            json.dump(cfg, f, separators=(",", ":\t"), indent=4, sort_keys=True)

    def log_tabular(self, key, val):
        if self.first_row:
            self.log_headers.append(key)
        else:
            assert key in self.log_headers, "Trying to introduce a new key %s that you didn't include in the first iteration" % key
        assert key not in self.log_current_row, "You already set %s this iteration. Maybe you forgot to call dump_tabular()" % key
        self.log_current_row[key] = val

    def dump_tabular(self):
        vals = [self.log_current_row.get(k, "") for k in self.log_headers]
        if not self.quiet:
            width = max(15, max((len(k) for k in self.log_headers), default=0))
            print("-" * (width + 22))
            for k, v in zip(self.log_headers, vals):
                print("| %*s | %15s |" % (width, k, ("%8.3g" % v) if hasattr(v, "__float__") else v))
            print("-" * (width + 22), flush=True)
        if self.first_row:
            self.output_file.write("\t".join(self.log_headers) + "\n")
        self.output_file.write("\t".join(map(str, vals)) + "\n")
        self.output_file.flush()
        self.log_current_row.clear()
        self.first_row = False


# ------------------------------------------------------------------------------------------------
# TensorBoard scalars without TensorFlow: what Actor.test() writes through tf.summary.FileWriter in the reference
# (algos/sac1/actor_learner.py:177-183,210-229: scalar "Reward" at step sample_times).  An event file is a TFRecord stream
# (length, masked crc32c of the length, payload, masked crc32c of the payload) of `Event` protobufs; the two messages needed
# are encoded by hand (Event{wall_time = 1: double, step = 2: int64, file_version = 3: string, summary = 5: Summary},
# Summary{value = 1: Value{tag = 1: string, simple_value = 2: float}}).  `tensorboard --logdir` reads the result.
# ------------------------------------------------------------------------------------------------
import struct

_CRC_TABLE = []


def _crc32c(data):
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


class SummaryWriter:
    """tf.summary.FileWriter(logdir) for scalars: add_scalar(tag, value, step) appends one event; flush() / close()."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), os.uname().nodename))
        self._f = open(self.path, "ab")
        self._write(struct.pack("<B", (1 << 3) | 1) + struct.pack("<d", time.time()) + _field_bytes(3, b"brain.Event:2"))

    def _write(self, event):
        header = struct.pack("<Q", len(event))
        self._f.write(header + struct.pack("<I", _masked_crc(header)) + event + struct.pack("<I", _masked_crc(event)))

    def add_scalar(self, tag, value, step):
        val = _field_bytes(1, tag.encode()) + struct.pack("<B", (2 << 3) | 5) + struct.pack("<f", float(value))
        summary = _field_bytes(1, val)
        event = (struct.pack("<B", (1 << 3) | 1) + struct.pack("<d", time.time()) + struct.pack("<B", (2 << 3) | 0) + _varint(int(step)) +
                 _field_bytes(5, summary))
        self._write(event)

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def read_scalars(path):
    """[(step, tag, value)] of an event file written by SummaryWriter (checks both checksums) — for tests."""
    out = []
    data = open(path, "rb").read()
    pos = 0
    while pos < len(data):
        (n,) = struct.unpack_from("<Q", data, pos)
        assert struct.unpack_from("<I", data, pos + 8)[0] == _masked_crc(data[pos:pos + 8])
        ev = data[pos + 12:pos + 12 + n]
        assert struct.unpack_from("<I", data, pos + 12 + n)[0] == _masked_crc(ev)
        pos += 12 + n + 4
        i, step, summary = 0, 0, None
        while i < len(ev):
            key = ev[i]; i += 1
            num, wt = key >> 3, key & 7
            if wt == 1:
                i += 8
            elif wt == 0:
                v, sh = 0, 0
                while True:
                    b = ev[i]; i += 1
                    v |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
                if num == 2:
                    step = v
            else:
                ln, sh = 0, 0
                while True:
                    b = ev[i]; i += 1
                    ln |= (b & 0x7F) << sh; sh += 7
                    if not b & 0x80:
                        break
                if num == 5:
                    summary = ev[i:i + ln]
                i += ln
        if summary is not None:   # Summary{value{tag, simple_value}}: one value per event here
            v = summary[2:] if summary[1] < 0x80 else summary[3:]
            tl = v[1]
            tag = v[2:2 + tl].decode()
            (val,) = struct.unpack_from("<f", v, 2 + tl + 1)
            out.append((step, tag, val))
    return out
