"""``snacc.cli`` (ref:snacc/cli.py:69,162,180; console script of ref:setup.py:115-117) -> :mod:`snacc_amd.cli`."""
from snacc_amd.cli import cli, log_template, tqdm_parallel_map  # noqa: F401

if __name__ == "__main__":
    cli()
