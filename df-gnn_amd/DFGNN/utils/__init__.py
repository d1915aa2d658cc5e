from .datasets import GraphDataLoader, load_data_full_graph, load_dataset_fn, mkdir  # noqa: F401
from .graph import Graph, batch  # noqa: F401
from .hipgraph import GraphedStep  # noqa: F401
from .util import *  # noqa: F401,F403
from .util import (Timer, benchmark, check_correct, inference_Graph_level, inference_Node_level,  # noqa: F401
                   parser_argument, preprocess_dglsp)
