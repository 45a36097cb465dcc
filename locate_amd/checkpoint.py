"""Checkpoints in the reference's on-disk format, plus the pieces the reference forgets.

The reference writes `torch.save(gen.state_dict(), OUTPUT_FOLDER/netG.torch)` and the same for `netD.torch` once per
pass of its outer loop (main.py:235-236) and has no load path.  `Generator.noise` - the constant [1, Z, 2, 2] map every
image is grown from - is a plain tensor (libs/models.py:59), so it is NOT in `state_dict()`: a reference checkpoint cannot
reproduce its own samples.  Here:

  * `netG.torch` / `netD.torch` hold exactly the reference's `state_dict()` (same keys, shapes, dtypes), so a file written
    by the reference loads here and a file written here loads into the reference (`load_state_dict(torch.load(...))`);
  * `netG.extra.torch` (a side file, so that `netG.torch` stays reference-compatible) holds `noise` and one flag: whether
    the discriminator's spectral-norm u/v are trainable yet (the reference's `dis.requires_grad_(True)` after the first
    G-step makes them so, main.py:172 - a piece of training state that `state_dict()` cannot carry);
  * `optG.torch` / `optD.torch` (optional) hold the Nadam states (`step`/`m_schedule` as float64, the two moments);
  * every load uses `weights_only=True`: nothing from a checkpoint file is ever executed.
"""
import os

import torch

G_FILE, D_FILE, G_EXTRA_FILE, G_OPT_FILE, D_OPT_FILE = "netG.torch", "netD.torch", "netG.extra.torch", "optG.torch", "optD.torch"


def _cpu(sd):
    return {k: (v.detach().to("cpu").clone() if torch.is_tensor(v) else v) for k, v in sd.items()}


def _cpu_tree(obj):
    if torch.is_tensor(obj):
        return obj.detach().to("cpu").clone()
    if isinstance(obj, dict):
        return {k: _cpu_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu_tree(v) for v in obj)
    return obj


def save_checkpoint(folder, gen, dis, gen_opt=None, dis_opt=None):
    """Writes netG.torch / netD.torch (reference layout, main.py:235-236), netG.extra.torch (the noise map) and, when
    optimizers are given, optG.torch / optD.torch.  Tensors are stored on the CPU.  Returns the list of files written."""
    os.makedirs(folder, exist_ok=True)
    written = []

    def put(name, obj):
        path = os.path.join(folder, name)
        tmp = path + ".tmp"
        torch.save(obj, tmp)
        os.replace(tmp, path)              # never leaves a half-written checkpoint under the final name
        written.append(path)

    if torch.cuda.is_available():
        torch.cuda.synchronize()           # the optimizer / graph replays write the weights through raw pointers
    put(G_FILE, _cpu(gen.state_dict()))
    put(D_FILE, _cpu(dis.state_dict()))
    uv = [p.requires_grad for n, p in dis.named_parameters() if n.endswith(("weight_u", "weight_v"))]
    put(G_EXTRA_FILE, {"noise": gen.noise.detach().to("cpu").clone(), "dis_uv_trainable": bool(uv) and all(uv)})
    if gen_opt is not None:
        put(G_OPT_FILE, _cpu_tree(gen_opt.state_dict()))
    if dis_opt is not None:
        put(D_OPT_FILE, _cpu_tree(dis_opt.state_dict()))
    return written


def _load(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def load_checkpoint(folder, gen, dis, gen_opt=None, dis_opt=None, strict=True):
    """Loads what save_checkpoint wrote - or what the reference wrote (then there is no netG.extra.torch: the noise map
    keeps its current value and `noise_restored` is False in the returned record).  Weight panels are re-packed lazily
    (load_state_dict bumps the parameters' version counters); optimizer files are optional."""
    rec = {"noise_restored": False, "gen_opt_restored": False, "dis_opt_restored": False, "dis_uv_trainable": False}
    gen.load_state_dict(_load(os.path.join(folder, G_FILE)), strict=strict)
    dis.load_state_dict(_load(os.path.join(folder, D_FILE)), strict=strict)
    extra = os.path.join(folder, G_EXTRA_FILE)
    if os.path.exists(extra):
        side = _load(extra)
        noise = side["noise"]
        if side.get("dis_uv_trainable", False):
            dis.requires_grad_(True)       # main.py:172 has already run in the checkpointed training state
            rec["dis_uv_trainable"] = True
        if tuple(noise.shape) != tuple(gen.noise.shape):
            raise ValueError("checkpoint noise map %s does not fit the generator's %s" % (tuple(noise.shape), tuple(gen.noise.shape)))
        with torch.no_grad():
            gen.noise.copy_(noise)         # in place: a captured hipGraph keeps reading the same address
        rec["noise_restored"] = True
    for opt, name, key in ((gen_opt, G_OPT_FILE, "gen_opt_restored"), (dis_opt, D_OPT_FILE, "dis_opt_restored")):
        path = os.path.join(folder, name)
        if opt is not None and os.path.exists(path):
            opt.load_state_dict(_load(path))
            rec[key] = True
    return rec
