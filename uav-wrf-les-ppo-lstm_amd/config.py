# config.py -- hyper-parameters of the PPOV2.0/2.1 trainer.
#
# Same module-level names and values as the reference's PPOV2.0/config.py:6-44 and
# PPOV2.1/config.py:12-13, imported by name by environment.py / model.py / train_ppo2.0.py
# (drop-in surface).  The block at the end is additive: knobs of the MI355X build
# whose defaults reproduce the reference (one env, MLP policy, 256-step buffer).

GRID_SIZE = 500
MAX_STEPS = 1000
CONC_PEAK = 100.0
TURBULENCE_INTENSITY = 3.0

# Gaussian field of PPOV2.1 (PPOV2.1/config.py:12-13)
GAUSSIAN_RADIUS = 15.0
PEAK_CONCENTRATION = 100.0

# PPO
GAMMA = 0.99
LAMBDA = 0.95
CLIP_EPSILON = 0.2
ENTROPY_BETA = 0.01
LEARNING_RATE = 3e-5
BATCH_SIZE = 256
EPOCHS = 5

# exploration
EXPLORE_BONUS = 0.6
DECAY_FACTOR = 0.999
GRID_DIVISIONS = 10
EXPLORE_DECAY_ALPHA = 0.002

# curriculum
INITIAL_RADIUS = 50.0
MIN_RADIUS = 5.0
RADIUS_DECAY = 0.9
SUCCESS_THRESHOLD = 0.6
WINDOW_SIZE = 120

# reward shaping
CONC_REWARD_COEF = 2.0
TKE_PENALTY_FACTOR = 0.4
BOUNDARY_PENALTY = 0.1
BOUNDARY_DECAY_START = 0.15

TRAINING_SIZE = 10
SUCCESS_DISTANCE_THRESHOLD = 40
EVALUATE_SIZE = 10

# ---------------------------------------------------------------------------- MI355X build (additive)
ENV_VARIANT = "v2.0"     # "v2.0" sigma=500/16 | "v2.1" sigma=GAUSSIAN_RADIUS | "v1.1" clip 500-1e-6, MAX_STEPS 5000
DEVICE = "cuda"
NUM_ENVS = 1             # vectorised trainer: environments per GPU (BASELINE C3: 4096)
HORIZON = BATCH_SIZE     # rollout length per env (BASELINE C3: 128)
POLICY = "mlp"           # "mlp" = the reference's PPOActorCritic | "lstm" = LSTM actor-critic (BASELINE)
HIDDEN = 128             # LSTM hidden size (64 / 128)
NUM_LAYERS = 1
NUM_MINIBATCHES = 1      # the reference uses ONE minibatch of the whole buffer per epoch (train_ppo2.0.py:44-45)
GAE_MODE = "reference_exact"   # or "standard" (PPOV1.0/ppo0.0.py:337-350)
SEED = 1234
TREND_K = 0              # extra observation channels obs[2](t) - obs[2](t-1-i), i < TREND_K (BASELINE C5: 2)
FIELD_BANK = None        # None = procedural fields | path to .npz / netCDF (conc, tke, source) | "synth:F" (BASELINE C4: "synth:64")
ITERATIONS = 200         # vectorised trainer: rollout + update iterations ...
EPISODES = None          # ... or stop once this many episodes have finished (the reference trains 2000, train_ppo2.0.py:128)
# data parallel, one process per GPU (torchrun / bench.py's launcher export these); 1 rank when absent
import os as _os
WORLD_SIZE = int(_os.environ.get("WORLD_SIZE", "1"))
RANK = int(_os.environ.get("RANK", "0"))
LOCAL_RANK = int(_os.environ.get("LOCAL_RANK", "0"))
DIST_BACKEND = _os.environ.get("UAVPPO_DIST_BACKEND", "nccl")   # "nccl" == RCCL over xGMI; "gloo" only for rehearsals / tests
