"""alphapig_amd: MI355X-native batched self-play / leaf-evaluation engine for the Gomoku
AlphaZero loop of anxingle/AlphaPig.

  game, game_ai, mcts_alphaZero, mcts_pure   drop-in host API of the reference
  policy_value_net                           PolicyValueNet on hand-written gfx950 HIP kernels
  selfplay                                   G concurrent games, one coalesced leaf batch per step
  dist                                       one process per GPU, all-gather of (s, pi, z)
"""
__version__ = "0.1.0"
