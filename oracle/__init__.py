"""TEST INFRASTRUCTURE ONLY -- not part of the product.

``oracle/`` holds a CPU (pure-torch fp32) restatement of the reference's
ToC3D / EVA-02 ViT backbone algorithm plus the harness that imports the real
reference (only inside the build container, where ``/root/reference`` exists)
to pin the restatement with golden vectors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import from here.  The product path
(``toc3d_amd``) never imports it and fails loudly if the HIP library is
missing -- there is no CPU fallback.
"""
