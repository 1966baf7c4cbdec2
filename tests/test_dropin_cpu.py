"""The reference drivers' own import lines resolve to ldmae_amd -- from a foreign working
directory, in fresh interpreters, in each of the three documented modes.  No GPU: only imports and module identity are checked."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ldmae_amd")

# /root/reference/LDMAE/train_accum.py:33-37 and inference.py:23-28, verbatim (minus what this tree does not mirror)
IMPORT_LINES = """
from models.lightningdit import LightningDiT_models
from transport import create_transport, Sampler
from datasets.img_latent_dataset import ImgLatentDataset
from tokenizer.util.misc import DiagonalGaussianDistribution
from tokenizer import models_mae
"""
CHECK = """
import sys, models.lightningdit, ldmae_amd.models.lightningdit, transport, ldmae_amd.transport, tokenizer.models_mae
assert models.lightningdit is ldmae_amd.models.lightningdit and transport is ldmae_amd.transport
assert models_mae is sys.modules["ldmae_amd.tokenizer.models_mae"]
assert callable(getattr(models_mae, "mae_for_ldmae_f8d16_prev"))
assert "LightningDiT-B/1" in LightningDiT_models and "LightningDiT-XL/1" in LightningDiT_models
t = create_transport("Linear", "velocity", None, None, None, use_cosine_loss=False, use_lognorm=True)
assert hasattr(Sampler(t), "sample_ode") and hasattr(t, "training_losses")
assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), "the product path must not import the oracle"
print("DROPIN-OK")
"""


def run(args, cwd, env_extra):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update(env_extra)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def fake_reference(tmp_path):
    """A directory shaped like /root/reference/LDMAE: its own models/ transport/ tokenizer/ packages (which must LOSE) plus a
    tokenizer submodule this tree does not mirror (which must still be served: inference.py:21 imports tokenizer.vavae)."""
    ref = tmp_path / "LDMAE"
    for pkg in ("models", "transport", "tokenizer", "datasets"):
        (ref / pkg).mkdir(parents=True)
        (ref / pkg / "__init__.py").write_text("WHO = 'reference'\n")
    (ref / "models" / "lightningdit.py").write_text("LightningDiT_models = 'reference'\n")
    (ref / "tokenizer" / "vavae.py").write_text("VA_VAE = 'reference-vavae'\n")
    (ref / "driver.py").write_text("from tokenizer.vavae import VA_VAE\nassert VA_VAE == 'reference-vavae'\n" + IMPORT_LINES + CHECK +
                                   "print('ARGV', sys.argv[1:])\n")
    return ref


def test_pythonpath_shadow_from_foreign_cwd(tmp_path):
    """PYTHONPATH=<repo>/ldmae_amd alone (the repo root is NOT on the path)."""
    out = run(["-c", IMPORT_LINES + CHECK], str(tmp_path), {"PYTHONPATH": PKG})
    assert "DROPIN-OK" in out


def test_package_names_still_work(tmp_path):
    code = "from ldmae_amd.models.lightningdit import LightningDiT_models\nfrom ldmae_amd.transport import create_transport\n" \
           "from ldmae_amd.tokenizer import models_mae\nfrom ldmae_amd.datasets.img_latent_dataset import ImgLatentDataset\nprint('PKG-OK')"
    assert "PKG-OK" in run(["-c", code], str(tmp_path), {"PYTHONPATH": ROOT})


def test_sitecustomize_hook_beats_the_script_directory(tmp_path):
    """`python driver.py` run from the reference directory (sys.path[0] = that directory, ahead of PYTHONPATH)."""
    ref = fake_reference(tmp_path)
    out = run(["driver.py", "--config", "x.yaml"], str(ref), {"PYTHONPATH": os.path.join(PKG, "dropin")})
    assert "DROPIN-OK" in out and "ARGV ['--config', 'x.yaml']" in out


def test_launcher_module(tmp_path):
    """python -m ldmae_amd.launch <driver.py> args..."""
    ref = fake_reference(tmp_path)
    out = run(["-m", "ldmae_amd.launch", "driver.py", "--config", "y.yaml"], str(ref), {"PYTHONPATH": ROOT})
    assert "DROPIN-OK" in out and "ARGV ['--config', 'y.yaml']" in out


def test_plain_run_of_the_fake_reference_is_the_reference(tmp_path):
    """Control: without any of the three modes the driver sees its own packages (so the tests above prove the redirection)."""
    ref = fake_reference(tmp_path)
    code = "from models.lightningdit import LightningDiT_models; print(LightningDiT_models)"
    assert "reference" in run(["-c", code], str(ref), {})


def test_under_accelerate_launch_the_reference_launcher(tmp_path):
    """run_train.sh:13-22 starts the driver with `accelerate launch ... train_accum.py`: the start-up hook must reach the rank process
    (PYTHONPATH is inherited), and `accelerate launch -m ldmae_amd.launch driver.py` must work too."""
    import shutil
    import pytest
    if shutil.which("accelerate") is None:
        pytest.skip("accelerate CLI not installed")
    ref = fake_reference(tmp_path)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for extra, args in (({"PYTHONPATH": os.path.join(PKG, "dropin")}, ["driver.py", "--config", "a.yaml"]),
                        ({"PYTHONPATH": ROOT}, ["-m", "ldmae_amd.launch", "driver.py", "--config", "a.yaml"])):
        r = subprocess.run(["accelerate", "launch", "--num_processes", "1", "--cpu"] + args, cwd=str(ref), env=dict(env, **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "DROPIN-OK" in r.stdout and "ARGV ['--config', 'a.yaml']" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
