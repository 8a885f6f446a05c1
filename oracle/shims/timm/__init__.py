"""Oracle shim (TEST INFRASTRUCTURE ONLY): CPU restatement of the timm 1.0.3 pieces that
/root/reference/videollama2/model/projector.py:22-23 imports.  timm is a third-party dependency of the
reference (pyproject.toml:24 pins timm==1.0.3) that is absent from /root/reference and from this image;
its published algorithm is restated here from SURVEY.md Appendix B.  PARITY UNPINNED against real timm."""
