"""Batched-graph inference harness -- same command line as the reference's DFGNN/script/test/test_batch_graph.py."""
import argparse

from DFGNN.script.harness import run_batch_graph
from DFGNN.utils import parser_argument

if __name__ == "__main__":
    run_batch_graph(parser_argument(argparse.ArgumentParser(description="batched-graph inference")))
