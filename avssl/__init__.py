"""Drop-in alias: `avssl.*` resolves to the MI355X implementation in `speechclip_amd`, so code written against the
reference package layout (run_task.py, example.py, checkpoints' pickled `avssl.base.OrderedNamespace`) runs unchanged."""
