#!/usr/bin/env python3
"""Same entry point as the reference: python infer.py -m rife -i IN -o OUT [-fps F | -t N] [-s] [-st T] [-hw] [-scale S].
The implementation lives in drba_amd/infer.py."""
from drba_amd.infer import inference, interpolate_stream, load_model, main, parse_args  # noqa: F401

if __name__ == "__main__":
    main()
