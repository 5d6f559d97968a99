"""The viewer's wire format: what run.py hands to the Dash thread after every
projection (src/web_vis.py:203-217, callers run.py:131, :206).

The reference's `update_dash_data(intersection_pcds, target_mesh)` turns the list of hit clouds
and the posed mesh into one dict of plain numpy arrays and puts it on a multiprocessing.Queue:

    {'pcds': [{'points': M x 3 float64, 'colors': M x 3 float64}, ...],
     'vertices': V x 3 float64, 'faces': F x 3 int32}

The Dash app itself (layout, callbacks, port 8050) is UI and out of scope; this module provides
the payload and the queue hand-over with the reference's names, so the loop of run.py runs
unchanged against the GPU path and any consumer of the queue sees the same message.
"""
import numpy as np

data_queue = None      # set by run_dash_app / attach_queues, like the module globals of web_vis.py:11-12
capture_queue = None


def attach_queues(data_q, capture_q=None):
    """What run_dash_app(data_q, capture_q) does before starting the server (web_vis.py:219-222)."""
    global data_queue, capture_queue
    data_queue, capture_queue = data_q, capture_q


class LatestQueue:
    """A data queue whose consumer keeps up: it holds the newest message only, like the Dash callback that drains
    `data_queue` on every tick and draws the last message it found (web_vis.py's interval callback).  What bench.py,
    the tools and the tests attach -- an ordinary queue.Queue that nobody reads keeps every frame's posed mesh alive,
    and with it the page-locked block its vertices were downloaded into (one hipHostMalloc per frame, 0.75 ms)."""

    def __init__(self):
        self.last, self.count = None, 0

    def put(self, item, *args, **kwargs):
        self.last = item
        self.count += 1

    def get(self, *args, **kwargs):
        item, self.last = self.last, None
        return item

    def empty(self):
        return self.last is None


def dash_payload(intersection_pcds, target_mesh):
    """The message of update_dash_data, built without sending it."""
    pcd_data = []
    for pcd in intersection_pcds:
        pcd_data.append({"points": np.asarray(pcd.points), "colors": np.asarray(pcd.colors)})
    return {"pcds": pcd_data, "vertices": np.asarray(target_mesh.vertices), "faces": np.asarray(target_mesh.triangles)}


def update_dash_data(intersection_pcds, target_mesh):
    """web_vis.py:203-217: build the message and put it on the data queue.  Returns the message
    (the reference returns None; nothing reads its return value)."""
    payload = dash_payload(intersection_pcds, target_mesh)
    if data_queue is None:
        raise RuntimeError("update_dash_data: no data queue attached (run_dash_app / attach_queues)")
    data_queue.put(payload)
    return payload
