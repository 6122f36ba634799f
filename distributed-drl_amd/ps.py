"""ParameterServer with the reference's surface (example/dsac.py:51-73; algos/sac1/sac1.py:66-100),
backed by one flat float32 device buffer (ddrl_ps_*).

The reference keeps `{var_name: ndarray}`; here the names map to (offset, shape) slices of the
flat buffer so that a push of the learner's whole parameter vector is ONE device copy (and ONE
RCCL broadcast across ranks, see comm.py).  Semantics kept: __init__/push snapshot by copy,
key-wise overwrite, pull(keys) returns the listed subset in order, get_weights returns the dict,
save_weights(name) pickles {key: ndarray} to name+"weights.pickle"."""
import ctypes
import pickle

import numpy as np
import torch

from . import _lib


class ParameterServer(object):
    def __init__(self, keys, values, weights_file="", device=None):
        _lib.require_gpu()
        self._lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if weights_file:
            # algos/sac1/sac1.py:72-81: restore from pickle, abort when the file is missing
            try:
                with open(weights_file, "rb") as pickle_in:
                    restored = pickle.load(pickle_in)
                    print("****** weights restored! ******")
            except Exception:
                print("------------------------------------------------")
                print(weights_file)
                print("------ error: weights file doesn't exist! ------")
                raise SystemExit()
            keys, values = list(restored.keys()), list(restored.values())
        self._table = {}
        self._order = []
        self._capacity = 0
        self._h = None
        self._extra = {}  # keys pushed later that do not fit the flat buffer's table
        self._extra_pushes = 0   # writes that never reach the device buffer count as versions too (RolloutDevice.pull compares versions)
        self._retired = {}       # key -> (offset, count, shape) of the flat run it held before a push of another shape moved it to _extra
        self.layout = 0          # bumped whenever a key enters or leaves the flat table: holders of a cached span() look again
        total = sum(int(np.asarray(v).size) if not torch.is_tensor(v) else int(v.numel()) for v in values)
        self._alloc(max(total, 1))
        off = 0
        for k, v in zip(keys, values):
            n = int(v.numel()) if torch.is_tensor(v) else int(np.asarray(v).size)
            shape = tuple(v.shape)
            self._table[k] = (off, n, shape)
            self._order.append(k)
            off += n
        self._used = off
        self.push(keys, values)
        self._version0 = self.version

    def _alloc(self, count):
        h = ctypes.c_void_p()
        _lib.check(self._lib.ddrl_ps_create(ctypes.byref(h), self.device.index, int(count)))
        self._h, self._capacity = h, int(count)
        p, c = ctypes.c_void_p(), ctypes.c_int64()
        _lib.check(self._lib.ddrl_ps_buffer(self._h, ctypes.byref(p), ctypes.byref(c)))
        from .replay import _view
        self.flat = _view(p.value, (int(c.value),), self.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ddrl_ps_destroy(h)

    def _dev(self, v):
        if torch.is_tensor(v):
            return v.to(device=self.device, dtype=torch.float32).contiguous().view(-1)
        return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(-1)).to(self.device)

    def push(self, keys, values):
        """Snapshot by copy, key-wise overwrite (example/dsac.py:59-62)."""
        s = _lib.stream_ptr()
        for k, v in zip(keys, values):
            if k not in self._table:
                # a new key: the reference just adds it to the dict
                self._extra[k] = self._dev(v).clone().view(tuple(v.shape))
                self._extra_pushes += 1
                if k not in self._order:
                    self._order.append(k)
                continue
            off, n, shape = self._table[k]
            src = self._dev(v)
            if src.numel() != n:
                # the reference's dict takes a value of any shape under an old key: the key leaves the flat buffer (its run stays
                # unused) and lives with the late-comers from here on
                self._retired[k] = self._table.pop(k)
                self._extra[k] = src.clone().view(tuple(v.shape))
                self._extra_pushes += 1
                self.layout += 1
                continue
            if tuple(v.shape) != shape:
                self._table[k] = (off, n, tuple(v.shape))
            _lib.check(self._lib.ddrl_ps_push(self._h, _lib.dptr(src), off, n, s))

    def push_flat(self, flat, offset=0):
        """One device copy for a contiguous run of parameters (the learner's whole vector)."""
        flat = self._dev(flat)
        _lib.check(self._lib.ddrl_ps_push(self._h, _lib.dptr(flat), int(offset), int(flat.numel()),
                                          _lib.stream_ptr()))
        # a key that had left the table for another shape: the flat push writes its old run, i.e. gives it a value of the old shape
        # again — key-wise overwrite, so the run is the key's value from here on (not the stale late-comer tensor)
        for k, (off, n, shape) in list(self._retired.items()):
            if int(offset) <= off and off + n <= int(offset) + int(flat.numel()):
                self._table[k] = (off, n, shape)
                del self._retired[k]
                self._extra.pop(k, None)
                self.layout += 1

    def pull(self, keys):
        """The listed subset, in the given order, as host float32 arrays (example/dsac.py:64-65;
        Ray serialises the return value, i.e. the caller gets copies)."""
        return [t.cpu().numpy() for t in self.pull_device(keys)]

    def pull_device(self, keys):
        out = []
        s = _lib.stream_ptr()
        for k in keys:
            if k in self._extra:
                out.append(self._extra[k].clone())
                continue
            off, n, shape = self._table[k]  # KeyError for unknown keys, like the reference's dict
            dst = torch.empty(n, dtype=torch.float32, device=self.device)
            _lib.check(self._lib.ddrl_ps_pull(self._h, _lib.dptr(dst), off, n, s))
            out.append(dst.view(shape))
        return out

    def pull_flat(self, offset, count, out=None):
        dst = out if out is not None else torch.empty(int(count), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.ddrl_ps_pull(self._h, _lib.dptr(dst), int(offset), int(count), _lib.stream_ptr()))
        return dst

    def span(self, keys):
        """(offset, count) when `keys` occupy one contiguous run of the flat buffer, else None."""
        if any(k not in self._table for k in keys):
            return None
        offs = sorted(self._table[k][:2] for k in keys)
        for (o0, n0), (o1, _) in zip(offs, offs[1:]):
            if o0 + n0 != o1:
                return None
        return offs[0][0], sum(n for _, n in offs)

    def get_weights(self):
        return {k: v for k, v in zip(self._order, self.pull(self._order))}

    # save weights to disk
    def save_weights(self, name=""):
        with open(name + "weights.pickle", "wb") as pickle_out:
            pickle.dump(self.get_weights(), pickle_out)

    @property
    def version(self):
        return int(self._lib.ddrl_ps_version(self._h)) + self._extra_pushes


class ParameterServerNode(ParameterServer):
    """The per-node flavour of algos/dqn/train.py:111-174: `ParameterServer(opt, weights_file, checkpoint_path,
    ps_index)` writes <save_dir>/All_Parameters.json, restores from <checkpoint_path>/checkpoint_weights.pickle
    when opt.recover (train.py:139-142) or from `weights_file` (:144-154), counts learner steps
    (`learner_step += opt.push_freq` per push, :164) and saves with save_weights() to
    <save_dir>/checkpoint/checkpoint_weights.pickle (:172-174).  keys/values default to a fresh Learner(opt)."""

    def __init__(self, opt, weights_file="", checkpoint_path="", ps_index=0, keys=None, values=None, device=None):
        import copy
        import json
        import os
        self.opt = opt
        self.learner_step = 0
        self.ps_index = ps_index
        if keys is None:
            from .agent import Learner
            keys, values = Learner(opt, job="ps").get_weights()
        os.makedirs(os.path.join(opt.save_dir, "checkpoint"), exist_ok=True)
        all_parameters = {k: v for k, v in copy.deepcopy(vars(opt)).items()}
        all_parameters["obs_space"] = ""
        all_parameters["act_space"] = ""
        with open(os.path.join(opt.save_dir, "All_Parameters.json"), "w") as fp:
            json.dump(all_parameters, fp, indent=4, sort_keys=True, default=str)
        if not checkpoint_path:
            checkpoint_path = os.path.join(opt.save_dir, "checkpoint")
        restored = None
        if getattr(opt, "recover", False):
            with open(os.path.join(checkpoint_path, "checkpoint_weights.pickle"), "rb") as pickle_in:
                restored = pickle.load(pickle_in)
                print("****** weights restored! ******")
        if restored is not None and not weights_file:
            keys, values = list(restored.keys()), list(restored.values())
        super().__init__(keys, values, weights_file=weights_file, device=device)
        self.learner_step = 0  # the constructor's initial fill is not a learner push

    def push(self, keys, values):
        super().push(keys, values)
        self.learner_step += self.opt.push_freq

    def save_weights(self):
        import os
        with open(os.path.join(self.opt.save_dir, "checkpoint", "checkpoint_weights.pickle"), "wb") as pickle_out:
            pickle.dump(self.get_weights(), pickle_out)
