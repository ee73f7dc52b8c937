"""Explicit, opt-in process-level runtime settings of the recorded pipelines.

Nothing here runs at import: a host application that embeds glenet_amd keeps the HIP runtime's defaults unless it asks.
"""
import os
import warnings

GRAPH_QUEUES_ENV = "DEBUG_HIP_FORCE_GRAPH_QUEUES"


def configure_graph_executor(queues=2):
    """ROCm 7.2's HIP-graph executor replays the branches of a recorded graph on `DEBUG_HIP_FORCE_GRAPH_QUEUES` streams
    (default 4).  The recorded GLENet-VR training step has three branches of which two are long: two executor queues
    replay it in 6.50 ms instead of 6.67 (profiles/r04_summary.md; 1 queue = no overlap 8.17 ms, 3: 6.63, 6 / 8: 6.67).

    The variable is a process-wide HIP *debug* switch that the runtime reads once, at its first call: it changes how
    EVERY HIP graph of the host process executes.  That is why this is a function the entry point calls (bench.py
    does, and echoes the value in its `config`), not something `import glenet_amd` does.  Call it before anything
    touches the GPU.  Returns the value in effect (a string) or None when the runtime's default holds.

    It must be the FIRST GPU-related call of the process: `torch.cuda.is_available()` already initialises HIP on ROCm while
    `torch.cuda.is_initialized()` stays False, so a call after it cannot be detected here -- `requested()` says what this
    module asked for, which equals what is in effect only under that condition.

    queues=None leaves the environment alone and only reports."""
    cur = os.environ.get(GRAPH_QUEUES_ENV)
    if queues is None or cur is not None:
        return cur                      # the caller's environment wins
    _requested[0] = str(int(queues))
    try:
        import torch
        started = torch.cuda.is_initialized()
    except Exception:                   # torch not importable yet: nothing has initialised HIP through it
        started = False
    if started:
        warnings.warn("glenet_amd.runtime.configure_graph_executor(%r) called after the HIP runtime was initialised: "
                      "%s is read at the first HIP call and would be ignored; recorded steps replay on the runtime's "
                      "default executor queues (measured +0.2 ms on the GLENet-VR step)" % (queues, GRAPH_QUEUES_ENV),
                      RuntimeWarning, stacklevel=2)
        return None
    os.environ[GRAPH_QUEUES_ENV] = str(int(queues))
    return os.environ[GRAPH_QUEUES_ENV]


_requested = [None]


def requested():
    """The value configure_graph_executor() asked for in this process (None: it never set anything).  It is what the runtime
    uses only if the call came before the first HIP call (see configure_graph_executor)."""
    return _requested[0]


def graph_executor_queues():
    """What a `config` block should echo: the executor-queue setting of this process ("default" = the runtime's own)."""
    return os.environ.get(GRAPH_QUEUES_ENV, "default")


_warned = [False]


def note_capture():
    """Called by the training pipelines when they record a step: one diagnostic (not an error) when the executor
    setting the headline was measured with is not in effect."""
    if _warned[0] or os.environ.get(GRAPH_QUEUES_ENV) is not None:
        return
    _warned[0] = True
    warnings.warn("recording a training step with the HIP runtime's default graph-executor queues; "
                  "glenet_amd.runtime.configure_graph_executor() before the first GPU call selects the two-queue "
                  "executor the published step time was measured with (opt-in, process-wide)", RuntimeWarning, stacklevel=3)
