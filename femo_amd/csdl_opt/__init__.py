"""Mirror of the reference's ``femo.csdl_opt`` package: the CSDL operator surface
(``FEAModel``, ``StateModel``/``StateOperation``, ``OutputModel``/``OutputOperation``)
on the HIP engine.  When ``csdl`` is importable its base classes are used;
otherwise the protocol stubs of ``_csdl_compat`` are, together with the in-repo
``Simulator`` that drives ``run()`` / ``compute_totals()`` the way the backend does.
"""
