"""Repository rules: the oracle is test infrastructure -- the product never imports it; no
reference source is vendored."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(os.path.join(ROOT, d)):
        for f in files:
            if f.endswith('.py'):
                yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for path in _py_files('speaker_follower_amd'):
        assert not pat.search(open(path).read()), '%s imports the oracle' % path


def test_only_allowed_files_import_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for name in os.listdir(ROOT):
        if name.endswith('.py') and name not in ('bench.py', '__graft_entry__.py'):
            assert not pat.search(open(os.path.join(ROOT, name)).read()), name


def test_runtime_files_do_not_read_the_reference():
    for path in list(_py_files('speaker_follower_amd')) + [os.path.join(ROOT, 'bench.py'),
                                                            os.path.join(ROOT, '__graft_entry__.py')]:
        assert '/root/reference' not in open(path).read(), path


def test_state_dict_key_contract():
    """Checkpoint contract (follower.py:1022-1035): key names of the four modules."""
    from speaker_follower_amd import model
    enc = model.EncoderLSTM(20, 12, 16, 0, 0.5)
    assert list(enc.state_dict()) == [
        'embedding.weight', 'lstm.weight_ih_l0', 'lstm.weight_hh_l0', 'lstm.bias_ih_l0',
        'lstm.bias_hh_l0', 'encoder2decoder.weight', 'encoder2decoder.bias']
    dec = model.AttnDecoderLSTM(24, 16, 0.5, feature_size=24)
    assert list(dec.state_dict()) == [
        'lstm.weight_ih', 'lstm.weight_hh', 'lstm.bias_ih', 'lstm.bias_hh',
        'visual_attention_layer.linear_in_h.weight', 'visual_attention_layer.linear_in_h.bias',
        'visual_attention_layer.linear_in_v.weight', 'visual_attention_layer.linear_in_v.bias',
        'text_attention_layer.linear_in.weight', 'text_attention_layer.linear_out.weight',
        'decoder2action.linear_in_h.weight', 'decoder2action.linear_in_h.bias',
        'decoder2action.linear_in_a.weight', 'decoder2action.linear_in_a.bias',
        'decoder2action.linear_out.weight', 'decoder2action.linear_out.bias']
    assert dec.u_begin.shape == (24,) and float(dec.u_begin.abs().sum()) == 0.0
    senc = model.SpeakerEncoderLSTM(24, 24, 16, 0.5)
    assert list(senc.state_dict()) == [
        'visual_attention_layer.linear_in_h.weight', 'visual_attention_layer.linear_in_h.bias',
        'visual_attention_layer.linear_in_v.weight', 'visual_attention_layer.linear_in_v.bias',
        'lstm.weight_ih', 'lstm.weight_hh', 'lstm.bias_ih', 'lstm.bias_hh',
        'encoder2decoder.weight', 'encoder2decoder.bias']
    sdec = model.SpeakerDecoderLSTM(20, 12, 16, 0.5)
    assert list(sdec.state_dict()) == [
        'embedding.weight', 'lstm.weight_ih', 'lstm.weight_hh', 'lstm.bias_ih', 'lstm.bias_hh',
        'attention_layer.linear_in.weight', 'attention_layer.linear_out.weight',
        'decoder2action.weight', 'decoder2action.bias']
