"""KITTI evaluation input (mirror of src/e2eflow/kitti/input.py for the parts Trainer.eval consumes)."""
