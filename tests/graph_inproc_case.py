"""In-process form of tests/graph_case.py for debugging (collected only when named on the command line):

    EGONET_AMD_GRAPH_MAX_N=16 python -m pytest tests/test_gpu_autograd.py tests/graph_inproc_case.py -q -m gpu
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('head', ['heatmap', 'coordinates'])
def test_graph_case_in_this_process(head):
    import graph_case
    old = os.environ.get('EGONET_AMD_GRAPH_MAX_N')
    try:
        graph_case.case(head)
    finally:
        if old is None:
            os.environ.pop('EGONET_AMD_GRAPH_MAX_N', None)
        else:
            os.environ['EGONET_AMD_GRAPH_MAX_N'] = old
