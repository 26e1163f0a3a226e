"""Drop-in entry point for cleanba/cleanba_ppo.py: `python -m cleanba_amd.cleanba_ppo --local-num-envs 120 ...`"""
from .args import parse_args
from .trainer import train


def main(argv=None):
    args = parse_args(argv, "ppo")
    return train(args, "ppo")


if __name__ == "__main__":
    main()
