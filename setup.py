"""Minimal packaging: keeps the reference's console-script name (ref:setup.py:115-117)."""
from setuptools import setup

setup(
    name="snacc_amd",
    version="0.1.0",
    packages=["snacc_amd", "snacc"],          # "snacc": the reference's import name, re-exporting snacc_amd
    package_data={"snacc_amd": ["libsnacc_hip.so", "csrc/*"]},
    entry_points={"console_scripts": ["snacc=snacc.cli:cli"]},
)
