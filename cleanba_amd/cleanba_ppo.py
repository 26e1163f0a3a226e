"""Drop-in entry point for cleanba/cleanba_ppo.py: `python -m cleanba_amd.cleanba_ppo --local-num-envs 120 ...`"""
from .args import parse_args
from .trainer import train


def main(argv=None):
    args = parse_args(argv, "ppo")
    from .launch import maybe_fan_out
    rc = maybe_fan_out(args, "cleanba_amd.cleanba_ppo", argv)   # split topologies: one worker process per GPU (cleanba_amd/launch.py)
    if rc is not None:
        raise SystemExit(rc)
    return train(args, "ppo")


if __name__ == "__main__":
    main()
