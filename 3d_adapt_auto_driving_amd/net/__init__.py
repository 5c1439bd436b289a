"""PointRCNN model assembly (inference branches of pointrcnn/lib/net and lib/rpn/proposal_layer.py)."""
