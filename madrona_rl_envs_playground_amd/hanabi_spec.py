"""Sizes of the Hanabi encodings as a function of the game configuration.

The reference asks DeepMind's hanabi_learning_environment for these
(envs/hanabi_env.py:76-77,92-96: ``vectorized_observation_shape()``,
``game.max_moves()``); that package is not a dependency here, so the sizes are
derived from the section layout of the simulator's own encoders
(/root/reference/src/hanabi_env/sim.hpp:13-30, sim.cpp:54-365).
"""

HAND_SIZE = 5          # players < 4 (sim.cpp:875)
NUM_PLAYERS = 2


def observation_size(config):
    k, r = int(config["colors"]), int(config["ranks"])
    bpc = k * r
    hands = HAND_SIZE * bpc * (NUM_PLAYERS - 1) + NUM_PLAYERS
    deck = (4 + (r - 2) * 2) * k - HAND_SIZE * NUM_PLAYERS
    board = deck + bpc + int(config["max_information_tokens"]) + int(config["max_life_tokens"])
    discards = 2 * r * k
    last_action = NUM_PLAYERS + 4 + NUM_PLAYERS + k + r + 2 * HAND_SIZE + bpc + 2
    knowledge = NUM_PLAYERS * HAND_SIZE * (bpc + k + r)
    return hands + board + discards + last_action + knowledge


def state_size(config):
    return observation_size(config) + int(config["colors"]) * int(config["ranks"]) * HAND_SIZE


def num_moves(config):
    return 2 * HAND_SIZE + (NUM_PLAYERS - 1) * int(config["colors"]) + (NUM_PLAYERS - 1) * int(config["ranks"])
