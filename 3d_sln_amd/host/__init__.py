"""Host-side mirror of the reference interfaces for the hot path.

Module names follow the reference files they stand in for:
  graph.py            <- models/graph.py
  Sg2ScVAE_model.py   <- models/Sg2ScVAE_model.py
  utils.py            <- utils.py (calculate_model_losses / add_loss)
  train.py            <- train.py (plus the 8-GPU data-parallel loop the reference lacks)
"""
