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
