"""`kge` for one rank of a multi-GPU job: `torchrun ... -m kge_amd.libkge_plugin.launch start cfg.yaml [kge options]`.

An unmodified `kge.cli.main()` runs underneath.  What N copies of `kge start` in one node cannot share, and what this
shim therefore sets per rank before handing over:
  * the output folder: `kge start` refuses a folder that exists (kge/cli.py:243-245), so N ranks racing for one name
    fail.  Rank 0 gets the folder the user named (`--folder`, or LibKGE's default local/experiments/<time>-<config>),
    rank r > 0 gets the sibling `<folder>-rank<r>` (its own log and trace; checkpoints are written by rank 0 only;
    a sub-folder would create <folder> under rank 0's feet);
  * `job.device`: the config file's `cuda` (or no setting) becomes `cuda:<LOCAL_RANK>`; a device named on the command
    line, or anything else in the config file (`cpu`, `cuda:3`), stands.
The process group (RCCL for cuda, gloo for cpu) is created here from torchrun's environment so that the folder name
can be agreed on; the hip_sharded_* jobs find it initialised.  Without torchrun's environment this is plain `kge`.
"""
import datetime
import os
import sys


def _arg(argv, name):
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return None


def _config_device(argv):
    """`job.device` of the config file named on the command line (flat `job.device:` or nested `job: {device: }`), or None."""
    cfg = next((a for a in argv[1:] if a.endswith((".yaml", ".yml")) and os.path.isfile(a)), None)
    if cfg is None:
        return None
    try:
        import yaml
        with open(cfg) as f:
            doc = yaml.safe_load(f) or {}
    except Exception:
        return None
    if isinstance(doc.get("job.device"), str):
        return doc["job.device"]
    job = doc.get("job")
    if isinstance(job, dict) and isinstance(job.get("device"), str):
        return job["device"]
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and "MASTER_ADDR" in os.environ:
        import torch
        import torch.distributed as dist
        device = on_cli = _arg(argv, "--job.device")
        if device is None:  # not on the command line: what the config file says (a YAML `job.device: cpu` must stand)
            device = _config_device(argv)
        cuda = torch.cuda.is_available() and (device is None or device.startswith("cuda"))
        # the config's bare `cuda` (LibKGE's default) becomes this rank's device; a device the command line names stands
        if cuda and on_cli is None and device in (None, "cuda"):
            argv += ["--job.device", f"cuda:{local}"]
        if cuda:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
        if argv and argv[0] == "start":
            folder = _arg(argv, "--folder")
            if folder is None:
                from kge.misc import kge_base_dir
                cfg = next((a for a in argv[1:] if a.endswith((".yaml", ".yml"))), "config.yaml")
                name = os.path.splitext(os.path.basename(cfg))[0]
                names = [os.path.join(kge_base_dir(), "local", "experiments",
                                      datetime.datetime.now().strftime("%Y%m%d-%H%M%S") + "-" + name)]
                dist.broadcast_object_list(names, src=0)  # rank 0's clock names the run
                folder = names[0]
                argv += ["--folder", folder]
            if rank > 0:
                i = next(k for k, a in enumerate(argv) if a == "--folder" or a.startswith("--folder="))
                mine = folder.rstrip("/") + f"-rank{rank}"
                if argv[i] == "--folder":
                    argv[i + 1] = mine
                else:
                    argv[i] = "--folder=" + mine
            os.makedirs(os.path.dirname(os.path.abspath(folder)) or ".", exist_ok=True)
    sys.argv = ["kge"] + argv
    from kge.cli import main as kge_main
    try:
        kge_main()
    finally:
        # every rank reaches the end before any rank closes its sockets (a rank tearing its communicator down while
        # another is inside its last collective aborts the straggler)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            try:
                dist.barrier()
            except Exception:
                pass
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
