"""Not collected by default (the file name): the graphed small-batch case IN the pytest process -- for the bisect of the
hipGraphLaunch crash (profiles/r5_graph_replay_crash.txt):  pytest tests/test_gpu_autograd.py tests/_graph_inproc.py"""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_graph_case_in_process(monkeypatch):
    import graph_case
    monkeypatch.setenv('EGONET_AMD_GRAPH_MAX_N', '16')
    graph_case.case('heatmap')
    os.environ['EGONET_AMD_GRAPH_MAX_N'] = '16'
