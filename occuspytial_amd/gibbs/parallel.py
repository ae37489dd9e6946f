"""Chain fan-out (reference ``occuspytial/gibbs/parallel.py:4-42``).

The reference copies the sampler once per extra chain and runs ``_run`` of every copy in its own
joblib process.  Here the copies are made the same way -- so chain ``k`` owns the generator the
reference would give it -- but a sampler that implements ``_run_chains`` receives all of them at
once and runs them as ONE batch on the GPU (chains are ``blockIdx.y`` of every kernel).
"""


def sample_parallel(sampler, **kwargs):
    chains = kwargs.pop('chains')
    samplers = [sampler] + [sampler.copy() for _ in range(chains - 1)]
    if hasattr(sampler, '_run_chains'):
        return sampler._run_chains(samplers, **kwargs)
    # generic samplers (a Python ``step``): one chain after the other on the host
    return [s._run(pos=pos, **kwargs) for pos, s in enumerate(samplers)]
