"""A scripted 2-player full-config Hanabi game worked out from the reference TEXT alone, for known-answer tests.

This is test infrastructure and deliberately NOT the oracle: nothing here imports, calls or mirrors oracle/hanabi_oracle.c.
It is a second, independent reading of /root/reference/src/hanabi_env/sim.cpp and rng.hpp, written as plain Python lists
of 0/1 in the order the reference appends them, for the move kinds and encoder sections the reference's own checker
(envs/hanabi_env.py:478-657) does not pin: the last-action section (sim.cpp:158-289), card knowledge (:291-331), the redraw /
shift-left of removeFromHand (:567-594), the deal (rng.hpp:7-36, drawDeck :45-52, deal order :508-519).

    game = Game(episode)            # world w of a fresh simulator plays episode w
    game.discard(slot) / game.play(slot) / game.hint_colour(c) / game.hint_rank(r)     # the player to move acts
    game.observation(observer)      # the 658 entries of generateObsState for `observer`, as the reference writes them
    game.own_hand(observer)         # the 125 entries encodeOwnHand appends to the state
    game.legal(observer)            # the 20 entries of generateActionMask

Information tokens may exceed their maximum: completing a firework adds one unconditionally (sim.cpp:676-678), the
reference then writes a longer thermometer with its running offset (:119-125) and every later section moves up; what no
longer fits the 658 / 783 entries of a row is cut off here (the reference writes it past the end of its array).  The game
is never over in a `script_*`; the endings -- checkDone (:812-850), the reward of a lost game, the re-deal of resetWorld with
the next episode index and BOTH agents' fresh rows -- are what `run_policy_games` plays through, many episodes per world.
"""
import numpy as np

K, R, HAND, MAX_INFO, MAX_LIFE = 5, 5, 5, 8, 3
ALL = (1 << (K * R)) - 1
PLAY, DISCARD, REVEAL_COLOUR, REVEAL_RANK, NONE = 0, 1, 2, 3, 4  # order of the move-type one-hot (sim.cpp:181-196)


def rng_seed(idx):
    """rng.hpp:7-26"""
    m = 0xFFFFFFFF
    v0, v1, s0 = idx & m, 0, 0
    for _ in range(8):
        s0 = (s0 + 0x9E3779B9) & m
        v0 = (v0 + (((((v1 << 4) & m) + 0xA341316C) & m) ^ ((v1 + s0) & m) ^ (((v1 >> 5) + 0xC8013EA4) & m))) & m
        v1 = (v1 + (((((v0 << 4) & m) + 0xAD90777D) & m) ^ ((v0 + s0) & m) ^ (((v0 >> 5) + 0x7E95761E) & m))) & m
    return v0


class Hand:
    """struct Hand (sim.hpp): five slots and a size; a slot behind `size` keeps whatever it held last."""

    def __init__(self):
        self.cards, self.plausible, self.known_colour, self.known_rank, self.size = [0] * HAND, [0] * HAND, [-1] * HAND, [-1] * HAND, 0


class Game:
    def __init__(self, episode):
        # resetWorld (sim.cpp:446-532): ordered deck of 3/2/2/2/1 copies per rank, five cards to player 0, then five to player 1
        self.deck = [R * c + r for c in range(K) for r in range(R) for _ in range(3 if r == 0 else 1 if r == R - 1 else 2)]
        self.generator = rng_seed(episode)
        self.hands = [Hand(), Hand()]
        for h in self.hands:
            for slot in range(HAND):
                h.cards[slot], h.plausible[slot] = self.draw(), ALL
            h.size = HAND
        self.fireworks = [0] * K
        self.discards = [0] * (K * R)
        self.info, self.life = MAX_INFO, MAX_LIFE
        self.mover, self.turns_to_play, self.score = 0, 2, 0
        self.last = dict(move=NONE, player=-1, target=-1, index=-1, colour=-1, rank=-1, reveal=0, scored=False, info_token=False)

    def draw(self):
        """drawDeck (sim.cpp:45-52): rand() = low 24 bits of an LCG step / 2^24 as float32 (rng.hpp:28-36); position =
        int(float32(size) * rand()); the drawn card is replaced by the last one."""
        self.generator = (1664525 * self.generator + 1013904223) & 0xFFFFFFFF
        rnd = np.float32(self.generator & 0xFFFFFF) / np.float32(0x1000000)
        at = int(np.float32(len(self.deck)) * rnd)
        card = self.deck[at]
        self.deck[at] = self.deck[-1]
        self.deck.pop()
        return card

    # ---- actionSystem (sim.cpp:596-792) ----
    def _begin(self, move):
        if not self.deck:
            self.turns_to_play -= 1
        who = self.mover
        self.last = dict(move=move, player=who, target=-1, index=-1, colour=-1, rank=-1, reveal=0, scored=False, info_token=False)
        self.mover = (who + 1) % 2
        return who, self.hands[who]

    def _remove(self, hand, slot):
        """removeFromHand (sim.cpp:567-594)"""
        if not self.deck:
            for i in range(slot + 1, hand.size):  # everything behind the slot moves down by one, the hand is one card shorter
                for field in (hand.cards, hand.plausible, hand.known_colour, hand.known_rank):
                    field[i - 1] = field[i]
            hand.size -= 1
        else:
            hand.cards[slot] = self.draw()
            hand.plausible[slot], hand.known_colour[slot], hand.known_rank[slot] = ALL, -1, -1

    def discard(self, slot):
        assert self.info < MAX_INFO, "script error: discarding needs a spent information token"
        _, hand = self._begin(DISCARD)
        card = hand.cards[slot]
        self.last.update(index=slot, colour=card // R, rank=card % R)
        self.discards[card] += 1
        self.info += 1
        self._remove(hand, slot)

    def play(self, slot):
        _, hand = self._begin(PLAY)
        card = hand.cards[slot]
        colour, rank = card // R, card % R
        self.last.update(index=slot, colour=colour, rank=rank)
        if self.fireworks[colour] == rank:
            self.fireworks[colour] += 1
            completed = self.fireworks[colour] == R
            self.info += 1 if completed else 0  # unconditionally: may exceed MAX_INFO (sim.cpp:676-678)
            self.last.update(scored=True, info_token=completed)
        else:
            self.discards[card] += 1
            self.life -= 1
        self._remove(hand, slot)

    def _hint(self, move, value):
        assert self.info > 0
        who, _ = self._begin(move)
        target = self.hands[1 - who]
        self.info -= 1
        by_colour = move == REVEAL_COLOUR
        named = sum(1 << (value * R + i if by_colour else i * R + value) for i in range(R))  # the five cards of the colour / rank
        reveal = 0
        for i in range(target.size):
            card = target.cards[i]
            if (card // R if by_colour else card % R) == value:
                reveal |= 1 << i
                if by_colour:
                    target.known_colour[i] = value
                else:
                    target.known_rank[i] = value
                target.plausible[i] &= named
            else:
                target.plausible[i] &= ~named
        assert reveal, "script error: the hint must touch a card"
        self.last.update(target=1 - who, reveal=reveal, **({"colour": value} if by_colour else {"rank": value}))

    def hint_colour(self, colour):
        self._hint(REVEAL_COLOUR, colour)

    def hint_rank(self, rank):
        self._hint(REVEAL_RANK, rank)

    def check_done(self):
        """checkDone (sim.cpp:812-850) -> (reward of both agents, game over): the score is the fireworks' sum, or NOTHING once the
        last life token is gone -- the move that loses the game is paid minus everything scored so far."""
        old = self.score
        self.score = sum(self.fireworks) if self.life > 0 else 0
        return self.score - old, self.life < 1 or self.score >= K * R or self.turns_to_play <= 0

    def action_id(self, kind, value):
        """uid of the reference's action enumeration: 0-4 discard slot, 5-9 play slot, 10-14 reveal colour, 15-19 reveal rank."""
        return {"discard": 0, "play": 5, "hint_colour": 10, "hint_rank": 15}[kind] + value

    # ---- generateObsState (sim.cpp:367-379), section by section, in the reference's order ----
    def observation(self, me):
        other = self.hands[1 - me]
        v = []
        # encodeHands (:54-90): the partner's cards, empty slots as zeros, then "hand is short" for me and for the partner
        for slot in range(HAND):
            v += [int(slot < other.size and b == other.cards[slot]) for b in range(K * R)]
        v += [int(self.hands[me].size < HAND), int(other.size < HAND)]
        # encodeBoard (:92-135)
        v += [int(i < len(self.deck)) for i in range(40)]
        for c in range(K):
            v += [int(i + 1 == self.fireworks[c]) for i in range(R)]
        v += [1] * self.info + [0] * max(0, MAX_INFO - self.info)  # two loops with a running offset (:119-125): longer when info > max
        v += [int(i < self.life) for i in range(MAX_LIFE)]
        # encodeDiscards (:137-156)
        for card in range(K * R):
            copies = 3 if card % R == 0 else 1 if card % R == R - 1 else 2
            v += [int(self.discards[card] > i) for i in range(copies)]
        # encodeLastAction (:158-289)
        lm = self.last
        relative = -1 if lm["player"] == -1 else (me - lm["player"] + 2) % 2
        v += [int(i == relative) for i in range(2)]
        v += [int(lm["move"] == kind) for kind in (PLAY, DISCARD, REVEAL_COLOUR, REVEAL_RANK)]
        hint = lm["move"] in (REVEAL_COLOUR, REVEAL_RANK)
        card_move = lm["move"] in (PLAY, DISCARD)
        v += [int(hint and i == (me - lm["target"] + 2) % 2) for i in range(2)]
        v += [int(lm["move"] == REVEAL_COLOUR and i == lm["colour"]) for i in range(K)]
        v += [int(lm["move"] == REVEAL_RANK and i == lm["rank"]) for i in range(R)]
        v += [int(hint and (lm["reveal"] >> i) & 1) for i in range(HAND)]
        v += [int(card_move and i == lm["index"]) for i in range(HAND)]
        v += [int(card_move and i == lm["colour"] * R + lm["rank"]) for i in range(K * R)]
        v += [int(lm["move"] == PLAY and lm["scored"]), int(lm["move"] == PLAY and lm["info_token"])]
        # encodeCardKnowledge (:291-331): my own cards, then the partner's.  Every one of the 25 plausibility entries of a
        # card repeats ONE bit of its mask -- bit <index of the player loop> (`1 << i`, :311), not bit <v>
        for i in range(2):
            hand = self.hands[(me + i) % 2]
            for slot in range(HAND):
                if slot < hand.size:
                    v += [(hand.plausible[slot] >> i) & 1] * (K * R)
                    v += [int(hand.known_colour[slot] == c) for c in range(K)]
                    v += [int(hand.known_rank[slot] == r) for r in range(R)]
                else:
                    v += [0] * (K * R + K + R)
        assert len(v) == 658 + max(0, self.info - MAX_INFO)
        return np.array(v, dtype=np.uint8)

    def state(self, me):
        """copyObsToState + encodeOwnHand (:333-365): the observation as long as it came out, then the own hand; 783 entries fit."""
        return np.concatenate([self.observation(me), self.own_hand(me)])[:783]

    def own_hand(self, me):
        """encodeOwnHand (:343-365): what the state carries behind the observation."""
        hand = self.hands[me]
        v = []
        for slot in range(HAND):
            v += [int(slot < hand.size and b == hand.cards[slot]) for b in range(K * R)]
        return np.array(v, dtype=np.uint8)

    def legal(self, me):
        """generateActionMask (:381-444); hint legality looks at all five card slots of the partner whatever the hand's size."""
        mine, other = self.hands[me], self.hands[1 - me]
        v = [int(i < mine.size and self.info < MAX_INFO) for i in range(HAND)]
        v += [int(i < mine.size) for i in range(HAND)]
        v += [int(self.info > 0 and any(card // R == c for card in other.cards)) for c in range(K)]  # all five slots (:416-417)
        v += [int(self.info > 0 and any(card % R == r for card in other.cards)) for r in range(R)]
        return np.array(v, dtype=np.int32)


def snapshot(g, uid, actor):
    """What the reference leaves visible after a move: the NEXT mover's fresh observation, state tail and legal moves."""
    me = g.mover
    return dict(uid=uid, actor=actor, mover=me, obs=g.observation(me)[:658], own=g.own_hand(me), state=g.state(me), legal=g.legal(me),
                info=g.info, life=g.life, deck=len(g.deck), scored=g.last["scored"], info_token=g.last["info_token"], move=g.last["move"])


def script_card_moves(episode):
    """Moves whose encodings the reference's checker skips, from the initial position: a colour hint, a rank hint, a play of a
    HINTED card (successful iff it is a rank-0 card), a discard of a hinted card, another play.  Returns the list of
    snapshots, one per move."""
    g = Game(episode)
    steps = []

    def did(kind, value):
        uid, actor = g.action_id(kind, value), g.mover
        getattr(g, kind)(value)
        steps.append(snapshot(g, uid, actor))

    did("hint_colour", g.hands[1].cards[0] // R)      # player 0 names the colour of player 1's first card
    did("hint_rank", g.hands[0].cards[2] % R)         # player 1 names the rank of player 0's third card
    did("play", 2)                                    # player 0 plays that very card: its slot is redrawn, knowledge reset
    did("discard", 0)                                 # player 1 discards its hinted first card
    did("play", 4)                                    # player 0 plays its last card
    return steps


def script_empty_deck(episode):
    """The whole deck: after two hints player 1 discards its first card on every turn and player 0 names that card's colour,
    until the 40th redraw empties the deck; then player 0 plays slot 1 WITHOUT a redraw -- the hand shifts left and is one
    card short (sim.cpp:573-582) -- and the game goes on for one more turn.  Returns one snapshot per move (83)."""
    g = Game(episode)
    steps = []

    def did(kind, value):
        uid, actor = g.action_id(kind, value), g.mover
        getattr(g, kind)(value)
        steps.append(snapshot(g, uid, actor))

    did("hint_colour", g.hands[1].cards[0] // R)
    did("hint_rank", g.hands[0].cards[3] % R)         # player 0's cards carry knowledge that has to move with them
    while g.deck:
        did("hint_colour", g.hands[1].cards[0] // R)
        did("discard", 0)
    assert g.mover == 0 and g.turns_to_play == 2 and len(steps) == 82
    did("play", 1)
    assert g.turns_to_play == 1 and g.hands[0].size == 4 and g.life >= 2
    return steps


def run_scripts(sim_step, read, scripts):
    """scripts[w] = snapshots of world w's game, or None; lengths may differ.  `sim_step(actions (2, n) int32)` advances the
    implementation under test, `read()` returns (obs (2, n, 658), state (2, n, 783), mask (2, n, 20), active (2, n), done (n,)).
    A world whose script is over (or that has none) goes on with the first legal move its own mask offers and is no longer
    looked at."""
    n = len(scripts)
    steps = max(len(sc) for sc in scripts if sc)
    kinds = set()
    for t in range(steps):
        acts = np.zeros((2, n), np.int32)
        filler = [w for w in range(n) if not scripts[w] or t >= len(scripts[w])]
        if filler:
            _, _, mask, active, _ = read()
            for w in filler:
                me = int(active[1, w] != 0)
                acts[me, w] = int(np.flatnonzero(mask[me, w])[0])
        for w in range(n):
            if scripts[w] and t < len(scripts[w]):
                acts[scripts[w][t]["actor"], w] = scripts[w][t]["uid"]
        sim_step(acts)
        obs, state, mask, active, done = read()
        for w in range(n):
            if not scripts[w] or t >= len(scripts[w]):
                continue
            snap = scripts[w][t]
            me = snap["mover"]
            assert not done[w], f"world {w}: the scripted game ended at move {t}"
            assert active[me, w] == 1 and active[1 - me, w] == 0
            got = obs[me, w].astype(np.uint8)
            if not np.array_equal(got, snap["obs"]):
                bad = np.flatnonzero(got != snap["obs"])
                raise AssertionError(f"world {w}, move {t} (action {snap['uid']}): observation entries {bad[:12].tolist()} differ from the by-hand answer")
            assert np.array_equal(state[me, w, :783].astype(np.uint8), snap["state"]), f"world {w}, move {t}: state (observation + own hand)"
            assert np.array_equal(mask[me, w], snap["legal"]), f"world {w}, move {t}: legal moves"
            kinds.add((snap["move"], bool(snap["scored"]), bool(snap["info_token"]), snap["info"] > MAX_INFO))
    return kinds


def script_complete_a_firework(episode, max_moves=140, after=4):
    """Both players play whatever is playable (the script knows every card), otherwise discard their first card when a token
    is missing, otherwise name the colour of the partner's first card -- until a firework is completed WHILE all eight
    information tokens are there: the ninth token (sim.cpp:676-678) lengthens the thermometer and moves every later section
    of the observation up by one.  `after` more moves follow under the same rule (a hint brings the count back to eight).
    Returns the snapshots, or None if this episode's deal does not get there."""
    g = Game(episode)
    steps, countdown = [], None
    for _ in range(max_moves):
        me = g.hands[g.mover]
        playable = [s for s in range(me.size) if g.fireworks[me.cards[s] // R] == me.cards[s] % R]
        if playable:
            kind, value = "play", max(playable, key=lambda s: me.cards[s] % R)
        elif g.info < MAX_INFO:
            kind, value = "discard", 0
        else:
            kind, value = "hint_colour", g.hands[1 - g.mover].cards[0] // R
        uid, actor = g.action_id(kind, value), g.mover
        getattr(g, kind)(value)
        if not g.deck or g.life < 1 or sum(g.fireworks) >= K * R:
            return None  # the endgame is not this script's subject
        steps.append(snapshot(g, uid, actor))
        if countdown is None and g.last["info_token"] and g.info > MAX_INFO:
            countdown = after
        elif countdown is not None:
            countdown -= 1
            if countdown == 0:
                return steps
    return None


# ---- whole games, endings included: every world follows a rule the script can evaluate (it sees all cards) ----
def policy_score_then_lose(g):
    """Play a playable card while nothing has been scored, then play cards that do not fit until the third life token is gone:
    games of four to eight moves that end with life_tokens < 1 and, usually, a reward of minus the score."""
    me = g.hands[g.mover]
    playable = [s for s in range(me.size) if g.fireworks[me.cards[s] // R] == me.cards[s] % R]
    wrong = [s for s in range(me.size) if s not in playable]
    if sum(g.fireworks) == 0 and playable:
        return "play", playable[0]
    return ("play", wrong[0]) if wrong else ("play", playable[0])


def policy_run_out_the_deck(g):
    """Discard the first card when a token is missing, otherwise name the colour of the partner's first card: forty redraws
    empty the deck, then every move counts turns_to_play down and the game ends with turns_to_play <= 0 and hands one card short."""
    if g.info < MAX_INFO:
        return "discard", 0
    return "hint_colour", g.hands[1 - g.mover].cards[0] // R


def run_policy_games(sim_step, read, n, policy, steps):
    """n worlds of a FRESH simulator (world w plays episode w, the next episode is n) follow `policy` for `steps` steps.
    `sim_step(actions (2, n) int32)`; `read()` -> (obs (2, n, 658), state (2, n, 783), mask (2, n, 20), active (2, n), done (n,),
    reward (2, n)).  After every step: reward and done of every world; for a world that goes on the mover's rows as in
    run_scripts; for a world whose game ended BOTH agents' rows of its next game, dealt from the episode index it gets
    when finished worlds take the indices in ascending world order.  Returns (episodes started, set of ending reasons)."""
    games = [Game(w) for w in range(n)]
    next_episode, reasons = n, set()
    for t in range(steps):
        acts = np.zeros((2, n), np.int32)
        for w, g in enumerate(games):
            kind, value = policy(g)
            acts[g.mover, w] = g.action_id(kind, value)
            getattr(g, kind)(value)
        sim_step(acts)
        obs, state, mask, active, done, reward = read()
        for w in range(n):  # ascending: the order in which finished worlds are given their episode indices
            g = games[w]
            rew, over = g.check_done()
            assert reward[0, w] == rew and reward[1, w] == rew, f"world {w}, step {t}: reward {reward[:, w]} against {rew} by hand"
            assert bool(done[w]) == over, f"world {w}, step {t}: done"
            if over:
                reasons.add("life" if g.life < 1 else "score" if g.score >= K * R else "turns")
                g = games[w] = Game(next_episode)
                next_episode += 1
                views = (0, 1)
            else:
                views = (g.mover,)
            assert active[g.mover, w] == 1 and active[1 - g.mover, w] == 0, f"world {w}, step {t}: active agent"
            for me in views:
                where = f"world {w}, step {t}, agent {me}" + (f" of the new game (episode {next_episode - 1})" if over else "")
                assert np.array_equal(obs[me, w].astype(np.uint8), g.observation(me)[:658]), where + ": observation"
                assert np.array_equal(state[me, w, :783].astype(np.uint8), g.state(me)), where + ": state"
                assert np.array_equal(mask[me, w], g.legal(me)), where + ": legal moves"
    return next_episode - n, reasons
