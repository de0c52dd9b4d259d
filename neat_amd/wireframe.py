"""Per-view 2-D wireframe container consumed inside forward (reference: code/datasets/utils/wireframe.py).
Same attributes and JSON keys (vertices, vertices-score, edges, edges-weights, height, width)."""
import json

import torch


class WireframeGraph:
    def __init__(self, vertices, v_confidences, edges, edge_weights, frame_width, frame_height):
        self.vertices = vertices.clone().detach()
        self.v_confidences = v_confidences.clone().detach()
        self.edges = edges.clone().detach()
        self.weights = edge_weights.clone().detach()
        self.frame_width, self.frame_height = frame_width, frame_height

    def line_segments(self, threshold=0.97):
        """[n,5] = (x1,y1,x2,y2,score) of the edges scoring above threshold."""
        keep = self.weights > threshold
        e = self.edges[keep]
        return torch.cat([self.vertices[e[:, 0]], self.vertices[e[:, 1]], self.weights[keep][:, None]], dim=-1)

    def rescale(self, image_width, image_height):
        self.vertices[:, 0] *= float(image_width) / float(self.frame_width)
        self.vertices[:, 1] *= float(image_height) / float(self.frame_height)
        self.frame_width, self.frame_height = image_width, image_height

    def jsonize(self):
        return {"vertices": self.vertices.cpu().tolist(), "vertices-score": self.v_confidences.cpu().tolist(),
                "edges": self.edges.cpu().tolist(), "edges-weights": self.weights.cpu().tolist(),
                "height": self.frame_height, "width": self.frame_width}

    @classmethod
    def load_json(cls, path):
        with open(path) as f:
            d = json.load(f)
        return cls(torch.tensor(d["vertices"], dtype=torch.float32), torch.tensor(d["vertices-score"], dtype=torch.float32),
                   torch.tensor(d["edges"], dtype=torch.long), torch.tensor(d["edges-weights"], dtype=torch.float32),
                   frame_width=d["width"], frame_height=d["height"])
