# Control-plane image (reference Docker/dockerfile: docker-ce + conda + Azure CLI + Jupyter on :9999, from which the
# notebooks build images, smoke-test them and submit jobs): docker CLI, ssh/rsync, Jupyter on :9999 and THIS repo's
# job tooling - tools/create_resources.py (cluster description + node prep), tools/local_smoke.py (what
# 00_CreateImageAndTest.ipynb does), tools/submit_job.py (01_Train*.ipynb), tools/prepare_imagenet.py
# (00_DataProcessing.ipynb) - installed as commands.  `make jupyter` mounts the checkout at /workspace and the
# docker socket, exactly like the reference's `make jupyter` (Makefile:38-46 upstream).
FROM python:3.10-slim
RUN apt-get update && apt-get install -y --no-install-recommends docker.io openssh-client rsync make git && \
    rm -rf /var/lib/apt/lists/* && pip install --no-cache-dir jupyter pyyaml python-dotenv
COPY deploy/docker/jupyter_notebook_config.py /root/.jupyter/
COPY tools /opt/ddl/tools
COPY deploy /opt/ddl/deploy
COPY Makefile /opt/ddl/Makefile
RUN for t in create_resources local_smoke submit_job prepare_imagenet; do \
        printf '#!/bin/bash\nexec python /opt/ddl/tools/%s.py "$@"\n' $t > /usr/local/bin/ddl-$t && chmod a+x /usr/local/bin/ddl-$t; \
    done
ENV PYTHONPATH=/workspace:/opt/ddl
WORKDIR /workspace
EXPOSE 9999
CMD ["bash"]
