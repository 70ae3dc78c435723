"""Action codec of the reference (azulnet/game_runner.py:102-111): a = display + 6*color + 30*pattern."""


def nn_serialize(display, color, pattern):
    return display + 6 * color + 30 * pattern


def nn_deserialize(i):
    i = int(i)
    return (i % 6, (i // 6) % 5, i // 30)
